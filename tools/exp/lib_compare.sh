#!/bin/bash
# bench.py's headline leg under several builds of the library on ONE box, alternating:  bash tools/exp/lib_compare.sh <rounds> lib1.so lib2.so ...
# ("default" = the in-tree library).  -> gpurun_out/lib_compare.json
set -u
cd "$(dirname "$0")/../.."
rounds=$1; shift
mkdir -p gpurun_out
python - "$rounds" "$@" <<'PY'
import json, os, subprocess, sys
rounds, libs = int(sys.argv[1]), sys.argv[2:]
rows = []
for i in range(rounds):
    for lib in libs:
        env = dict(os.environ)
        env.pop('DAAM_HIP_LIB', None)
        if lib != 'default':
            env['DAAM_HIP_LIB'] = lib
        p = subprocess.run([sys.executable, 'bench.py', '--no-baselines', '--no-integrated', '--no-pmc', '--no-other-configs', '--steps', '30', '--warmup', '5'], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        r = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
        row = dict(lib=os.path.basename(lib), maps_per_s=r['value'], tap_ms=r['roofline']['ms_per_launch'],
                   mhz=(r['roofline_issue'] or {}).get('clock', {}).get('mhz_median_under_load'), fin_us=round(r['roofline_finalize']['ms_per_launch'] * 1e3, 1))
        rows.append(row)
        print(row, flush=True)
json.dump(rows, open('gpurun_out/lib_compare.json', 'w'), indent=1)
PY

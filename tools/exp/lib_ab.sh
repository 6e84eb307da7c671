#!/bin/bash
# A / B of two builds of the library on ONE box: bench.py's headline leg under $1 (a variant .so) and under the default library, alternating.
#   gpurun -- 'bash tools/exp/lib_ab.sh tools/exp/libdaam_prev.so [rounds] [bench args]'  ->  gpurun_out/lib_ab.json
set -u
cd "$(dirname "$0")/../.."
other=$1; rounds=${2:-3}; shift; shift || true
mkdir -p gpurun_out
python - "$other" "$rounds" "$@" <<'PY'
import json, os, subprocess, sys
other, rounds, extra = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
rows = []
for i in range(rounds):
    for tag, lib in (('other', other), ('default', None)):
        env = dict(os.environ)
        if lib:
            env['DAAM_HIP_LIB'] = lib
        else:
            env.pop('DAAM_HIP_LIB', None)
        p = subprocess.run([sys.executable, 'bench.py', '--no-baselines', '--no-integrated', '--no-pmc', '--no-other-configs', *extra], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        r = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
        row = dict(lib=tag, maps_per_s=r['value'], tap_ms=r['roofline']['ms_per_launch'], hbm_frac=r['roofline']['frac'],
                   mhz=(r['roofline_issue'] or {}).get('clock', {}).get('mhz_median_under_load'), fin_us=round(r['roofline_finalize']['ms_per_launch'] * 1e3, 1))
        rows.append(row)
        print(row, flush=True)
json.dump(dict(other=other, rows=rows), open('gpurun_out/lib_ab.json', 'w'), indent=1)
PY

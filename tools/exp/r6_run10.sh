#!/bin/bash
# round 6, GPU call 10: SD-v1.5 slab launch, every permutation of the three class segments (the half-size tail last) against the shipped order
# (160, 40, 80); variant libraries = daam_api.hip with another seg_rank (host-side table order only: same kernel, same sums).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 4080160 8040160; do
  DAAM_HIP_LIB=tools/exp/libdaam_ord$v.so timeout 600 python -m pytest tests/test_gpu_slab.py -q -m gpu -x -k "bit_identical_to_chunked" 2>&1 | tail -1
done
python - <<'PY'
import json, os, subprocess, sys
def bench(args, env_extra):
    env = dict(os.environ, BENCH_FULL_RECORD='/tmp/bench_full_ab.json', **env_extra)
    p = subprocess.run([sys.executable, 'bench.py', '--no-baselines', '--no-integrated', '--no-pmc', '--no-other-configs', '--no-sustained', *args], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    if p.returncode: print(p.stderr[-2000:])
    r = json.load(open('/tmp/bench_full_ab.json'))
    return dict(maps_per_s=r['value'], tap_ms=r['roofline']['ms_per_launch'], tap_ms_iso=r['roofline']['ms_per_launch_isolated'])
rows = []
for i in range(3):
    for tag in ('default', '1608040', '4016080', '4080160', '8016040', '8040160'):
        env = {} if tag == 'default' else dict(DAAM_HIP_LIB=f'tools/exp/libdaam_ord{tag}.so')
        row = dict(order='160,40,80 (shipped)' if tag == 'default' else tag, **bench(['--workload', 'sd15', '--steps', '100', '--warmup', '10'], env)); rows.append(row); print(row, flush=True)
json.dump(rows, open('gpurun_out/r6_run10_ab.json', 'w'), indent=1)
PY

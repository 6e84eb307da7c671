#!/bin/bash
# round 5, fourth GPU call: half-size tail workgroups of the slab kernel -- parity, then a sweep of DAAM_SLAB_TAIL on the SD-v1.5 workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_slab.py tests/test_gpu_parity.py -x -q -m gpu -k "slab or eight_wave" 2>&1 | tail -4
A="--workload sd15 --no-baselines --no-integrated --no-pmc --no-sustained --steps 100 --warmup 10"
for rep in 1 2; do
  for t in 0 25 50 60 75 100; do
    DAAM_SLAB_TAIL=$t timeout 200 python bench.py $A > gpurun_out/r5_4_tail_${t}_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5_4_tail_*.json')):
    try:
        r=json.load(open(f)); ro=r['roofline']
        print(f, r['value'], 'tap ms region', ro['ms_per_launch'], 'iso', ro['ms_per_launch_isolated'], 'frac', ro['frac'])
    except Exception as e:
        print(f, 'ERR', e)
PY

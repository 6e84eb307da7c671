#!/bin/bash
# round 5: head-minor workgroup numbering of the head_dim-64 deferred launches (DAAM_TAP_HEAD_MINOR=1 / 0), alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
A="--no-baselines --no-integrated --no-other-configs --no-pmc --no-sustained --warmup 10 --steps 100"
for rep in 1 2 3; do
  for hm in 0 1; do
    DAAM_TAP_HEAD_MINOR=$hm timeout 200 python bench.py $A 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('head_minor=$hm sdxl1024', r['value'], r['roofline']['ms_per_launch'], r['roofline']['ms_per_launch_isolated'])"
  done
done
DAAM_TAP_HEAD_MINOR=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -k "eight_wave or full_size or deferred_equals" 2>&1 | tail -2

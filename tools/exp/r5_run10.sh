#!/bin/bash
# round 5: non-temporal Q fetches (-DDAAM_TAP_Q_AUX=2 -DDAAM_SLAB_Q_AUX=2) against the default policy, headline and SD-v1.5, alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
A="--no-baselines --no-integrated --no-other-configs --no-pmc --no-sustained --warmup 10"
for rep in 1 2 3; do
  for lib in default tools/exp/libdaam_qnt.so; do
    tag=$(basename $lib .so)
    if [ $lib = default ]; then unset DAAM_HIP_LIB; else export DAAM_HIP_LIB=$lib; fi
    timeout 200 python bench.py $A --steps 100 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$tag sdxl1024', r['value'], r['roofline']['ms_per_launch'], r['roofline']['ms_per_launch_isolated'])"
    timeout 200 python bench.py $A --steps 100 --workload sd15 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('$tag sd15', r['value'], r['roofline']['ms_per_launch'], r['roofline']['ms_per_launch_isolated'])"
  done
done
unset DAAM_HIP_LIB
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "failed_announcement" 2>&1 | tail -2

#!/bin/bash
# round 5, first GPU call: slab kernel parity, then SD-v1.5 A/B (slab vs chunked) on one box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_slab.py tests/test_gpu_chunked.py -x -q -m gpu > gpurun_out/r5_1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5_1_tests.log
tail -15 gpurun_out/r5_1_tests.log
for rep in 1 2; do
  timeout 400 python bench.py --workload sd15 --steps 20 --warmup 5 --no-baselines --no-integrated $([ $rep = 2 ] && echo --no-pmc) > gpurun_out/r5_1_sd15_slab_$rep.json 2> gpurun_out/r5_1_sd15_slab_$rep.err
  DAAM_TAP_SLAB=0 timeout 400 python bench.py --workload sd15 --steps 20 --warmup 5 --no-baselines --no-integrated --no-pmc > gpurun_out/r5_1_sd15_chunk_$rep.json 2> gpurun_out/r5_1_sd15_chunk_$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5_1_sd15_*.json')):
    try:
        r=json.load(open(f)); ro=r['roofline']
        print(f, r['value'], 'tap ms', ro['ms_per_launch'], 'iso', ro['ms_per_launch_isolated'], 'frac', ro['frac'], 'traffic', ro.get('traffic'), ro.get('traffic_over_algorithmic'), ro.get('traffic_in_run_note'))
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 600 python -m pytest tests/test_gpu_integration.py -x -q -m gpu -k "sd15_full_stack" > gpurun_out/r5_1_integ.log 2>&1; tail -5 gpurun_out/r5_1_integ.log

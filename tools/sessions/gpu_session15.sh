#!/bin/bash
mkdir -p gpurun_out/s15
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "tap_qk or deferred or wide or generic" > gpurun_out/s15/pytest.log 2>&1
DAAM_HIP_LIB=$R/build/libdaam_tap_q2.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "tap_qk or deferred or wide" > gpurun_out/s15/pytest_q2.log 2>&1
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-baselines --no-integrated > gpurun_out/s15/bench_$i.json 2> gpurun_out/s15/bench_$i.err
DAAM_HIP_LIB=$R/build/libdaam_tap_q2.so timeout 300 python bench.py --steps 10 --warmup 3 --no-baselines --no-integrated > gpurun_out/s15/bench_q2_$i.json 2> gpurun_out/s15/bench_q2_$i.err
done
tail -2 gpurun_out/s15/pytest.log; tail -2 gpurun_out/s15/pytest_q2.log
python -c "
import json
for n in ('bench_1','bench_q2_1','bench_2','bench_q2_2'):
    try:
        d=json.load(open('gpurun_out/s15/%s.json'%n)); print(n, d['value'], 'tap', d['roofline']['ms_per_launch'], 'clock', d['roofline_issue']['clock']['mhz_median_under_load'])
    except Exception as e: print(n, 'ERR', e)
"

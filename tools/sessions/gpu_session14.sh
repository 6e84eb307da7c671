#!/bin/bash
mkdir -p gpurun_out/s14
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "tap_qk or deferred or golden or wide or generic" > gpurun_out/s14/pytest.log 2>&1
for i in 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-baselines --no-integrated > gpurun_out/s14/bench_$i.json 2> gpurun_out/s14/bench_$i.err
DAAM_HIP_LIB=$R/build/libdaam_tap_pf.so timeout 300 python bench.py --steps 10 --warmup 3 --no-baselines --no-integrated > gpurun_out/s14/bench_pf_$i.json 2> gpurun_out/s14/bench_pf_$i.err
done
DAAM_HIP_LIB=$R/build/libdaam_tap_pf.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "tap_qk or deferred" > gpurun_out/s14/pytest_pf.log 2>&1
tail -2 gpurun_out/s14/pytest.log; tail -2 gpurun_out/s14/pytest_pf.log
python -c "
import json
for n in ('bench_1','bench_pf_1','bench_2','bench_pf_2'):
    try:
        d=json.load(open('gpurun_out/s14/%s.json'%n)); print(n, d['value'], 'tap', d['roofline']['ms_per_launch'], 'clock', d['roofline_issue']['clock']['mhz_median_under_load'])
    except Exception as e: print(n, 'ERR', e)
"

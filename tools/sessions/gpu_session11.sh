#!/bin/bash
mkdir -p gpurun_out/s11
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -k "finalize or golden or full_size or properties or tap_qk" > gpurun_out/s11/pytest.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-baselines --no-integrated > gpurun_out/s11/bench.json 2> gpurun_out/s11/bench.err
DAAM_HIP_LIB=$R/build/libdaam_tap_prio.so timeout 300 python bench.py --steps 20 --warmup 5 --no-baselines --no-integrated > gpurun_out/s11/bench_prio.json 2> gpurun_out/s11/bench_prio.err
timeout 300 python bench.py --steps 5 --warmup 2 --workload sdxl2048 --denoise-steps 100 --no-baselines > gpurun_out/s11/bench_sdxl2048.json 2> gpurun_out/s11/bench_sdxl2048.err
tail -3 gpurun_out/s11/pytest.log
python -c "
import json
for n in ('bench','bench_prio','bench_sdxl2048'):
    try:
        d=json.load(open('gpurun_out/s11/%s.json'%n)); print(n, d['value'], 'tap', d['roofline']['ms_per_launch'], d['roofline']['frac'], 'fin', d['roofline_finalize']['ms_per_launch'], d['roofline_finalize']['frac'])
    except Exception as e: print(n, 'ERR', e)
"

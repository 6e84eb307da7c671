#!/bin/bash
mkdir -p gpurun_out/s9
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -k "finalize or golden or two_resolutions or full_size" > gpurun_out/s9/pytest.log 2>&1
timeout 100 python tools/fin_sweep.py > gpurun_out/s9/sweep.txt 2>&1
DAAM_FIN_UNIFORM=1 timeout 100 python tools/fin_sweep.py >> gpurun_out/s9/sweep.txt 2>&1
DAAM_HIP_LIB=$R/build/libdaam_fin_timing.so DAAM_NO_PAIRED_FINALIZE=1 timeout 120 python tools/fin_timing.py > gpurun_out/s9/fin_timing_unpaired.txt 2>&1
DAAM_HIP_LIB=$R/build/libdaam_fin_timing.so timeout 120 python tools/fin_timing.py > gpurun_out/s9/fin_timing_paired.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 --no-baselines --no-integrated > gpurun_out/s9/bench.json 2> gpurun_out/s9/bench.err
DAAM_HIP_LIB=$R/build/libdaam_tap_pf.so timeout 300 python bench.py --steps 20 --warmup 5 --no-baselines --no-integrated > gpurun_out/s9/bench_pf.json 2> gpurun_out/s9/bench_pf.err
tail -3 gpurun_out/s9/pytest.log; grep -h finalize_us gpurun_out/s9/sweep.txt | cut -c1-150; tail -9 gpurun_out/s9/fin_timing_unpaired.txt | head -6
python -c "
import json
for n in ('bench','bench_pf'):
    d=json.load(open('gpurun_out/s9/%s.json'%n)); print(n, 'tap', d['roofline']['ms_per_launch'], 'fin', d['roofline_finalize']['ms_per_launch'], d['value'])
"

#!/bin/bash
mkdir -p gpurun_out/s4
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_processor.py -m gpu -q --timeout 600 > gpurun_out/s4/pytest.log 2>&1
DAAM_HIP_LIB=$R/build/libdaam_fin_timing.so DAAM_NO_PAIRED_FINALIZE=1 timeout 120 python tools/fin_timing.py > gpurun_out/s4/fin_timing_unpaired.txt 2>&1
DAAM_HIP_LIB=$R/build/libdaam_fin_timing.so timeout 120 python tools/fin_timing.py > gpurun_out/s4/fin_timing_paired.txt 2>&1
DAAM_HIP_LIB=$R/build/libdaam_fin_abl1.so timeout 120 python bench.py --steps 10 --warmup 3 --no-baselines > gpurun_out/s4/bench_abl1.json 2>gpurun_out/s4/bench_abl1.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s4/prof_paired -- python $R/bench.py --no-baselines --steps 20 --warmup 2 > $R/gpurun_out/s4/prof_paired.log 2>&1
DAAM_NO_PAIRED_FINALIZE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s4/prof_unpaired -- python $R/bench.py --no-baselines --steps 20 --warmup 2 > $R/gpurun_out/s4/prof_unpaired.log 2>&1
cd $R
for d in prof_paired prof_unpaired; do f=$(find gpurun_out/s4/$d -name '*kernel_stats.csv' | head -1); echo "== $d"; head -8 "$f" | cut -c1-200; cp "$f" gpurun_out/s4/${d}_kernel_stats.csv; rm -rf gpurun_out/s4/$d; done > gpurun_out/s4/stats_head.txt 2>&1
grep -E "passed|failed" gpurun_out/s4/pytest.log | tail -2

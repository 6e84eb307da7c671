#!/bin/bash
mkdir -p gpurun_out/s5
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
timeout 120 tools/ubench_fin > gpurun_out/s5/ubench_fin.txt 2>&1
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/s5/bench.json 2> gpurun_out/s5/bench.err
timeout 600 python -m pytest tests/test_gpu_processor.py tests/test_gpu_distributed.py -m gpu -q --timeout 600 > gpurun_out/s5/pytest.log 2>&1
rocprofv3 -L 2>/dev/null | grep -iE "MFMA|SQ_INSTS_VALU|SQ_ACTIVE_INST|SQ_BUSY_CY|SQ_WAVE_CYCLES|SQ_WAIT" | head -60 > gpurun_out/s5/counters.txt
cat gpurun_out/s5/ubench_fin.txt; tail -3 gpurun_out/s5/pytest.log; tail -c 600 gpurun_out/s5/bench.err

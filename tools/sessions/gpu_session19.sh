#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/s19
timeout 600 python -m pytest tests/test_gpu_attend.py -m gpu -q -x --timeout 300 > gpurun_out/s19/attend.log 2>&1
tail -n 25 gpurun_out/s19/attend.log
timeout 900 python -m pytest tests/test_gpu_processor.py tests/test_gpu_integration.py -m gpu -q --timeout 600 > gpurun_out/s19/proc.log 2>&1
tail -n 15 gpurun_out/s19/proc.log
timeout 300 python tools/overhead_probe.py 50 9 > gpurun_out/s19/probe50.json 2> gpurun_out/s19/probe50.err
cat gpurun_out/s19/probe50.json

#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/s18
timeout 300 python tools/overhead_probe.py 50 9 > gpurun_out/s18/probe50c.json 2> gpurun_out/s18/probe50c.err
DAAM_SYNC_RELEASE=1 timeout 300 python tools/overhead_probe.py 50 9 > gpurun_out/s18/probe50c_sync.json 2> gpurun_out/s18/probe50c_sync.err
timeout 600 python -m pytest tests/test_gpu_integration.py tests/test_gpu_processor.py -m gpu -q -x --timeout 600 > gpurun_out/s18/pytest.log 2>&1
cat gpurun_out/s18/probe50c.json gpurun_out/s18/probe50c_sync.json; tail -n 3 gpurun_out/s18/probe50c.err gpurun_out/s18/pytest.log

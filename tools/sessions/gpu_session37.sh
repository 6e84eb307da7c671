#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/s37
timeout 300 python -m pytest tests/test_gpu_attend.py -m gpu -q --timeout 300 > gpurun_out/s37/pytest.log 2>&1
tail -n 5 gpurun_out/s37/pytest.log

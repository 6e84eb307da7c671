#!/bin/bash
mkdir -p gpurun_out/s2
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python tools/debug_ulps.py > gpurun_out/s2/debug_ulps.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s2/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s2/pytest.log
grep -E "passed|failed" gpurun_out/s2/pytest.log | tail -3

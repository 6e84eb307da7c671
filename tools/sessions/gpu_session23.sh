#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/s23; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_attend.py -m gpu -q --timeout 300 > $O/attend.log 2>&1
tail -n 30 $O/attend.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_processor.py -m gpu -q -x --timeout 600 > $O/parity.log 2>&1
tail -n 12 $O/parity.log

#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/s33; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_attend.py tests/test_gpu_processor.py tests/test_gpu_integration.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1
tail -n 12 $O/pytest.log
timeout 300 python tools/overhead_probe.py 50 9 > $O/probe50.json 2> $O/probe50.err
cat $O/probe50.json; cat gpurun_out/integrated_overhead.json

#!/bin/bash
mkdir -p gpurun_out/s7
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
timeout 900 python -m pytest tests/test_gpu_evaluate.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "evaluate or iou or overlap or finalize or golden or word" > gpurun_out/s7/pytest.log 2>&1
for c in default 8 10 16 20; do
  if [ $c = default ]; then timeout 100 python tools/fin_sweep.py; else DAAM_FIN_CHUNKS=$c timeout 100 python tools/fin_sweep.py; fi
done > gpurun_out/s7/sweep_chunks.txt 2>&1
DAAM_NO_PARTIAL_FINALIZE=1 timeout 100 python tools/fin_sweep.py >> gpurun_out/s7/sweep_chunks.txt 2>&1
tail -3 gpurun_out/s7/pytest.log; grep -h finalize_us gpurun_out/s7/*.txt

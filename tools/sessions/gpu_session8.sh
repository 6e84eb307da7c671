#!/bin/bash
mkdir -p gpurun_out/s8
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 -k "finalize or golden or two_resolutions or full_size" > gpurun_out/s8/pytest.log 2>&1
for c in default 8 10; do
  if [ $c = default ]; then timeout 100 python tools/fin_sweep.py; else DAAM_FIN_CHUNKS=$c timeout 100 python tools/fin_sweep.py; fi
done > gpurun_out/s8/sweep.txt 2>&1
DAAM_NO_PARTIAL_FINALIZE=1 timeout 100 python tools/fin_sweep.py >> gpurun_out/s8/sweep.txt 2>&1
DAAM_NO_PAIRED_FINALIZE=1 timeout 100 python tools/fin_sweep.py >> gpurun_out/s8/sweep.txt 2>&1
for l in k3w4 k3w3; do DAAM_HIP_LIB=$R/build/libdaam_fin_$l.so timeout 100 python tools/fin_sweep.py; done >> gpurun_out/s8/sweep.txt 2>&1
DAAM_HIP_LIB=$R/build/libdaam_fin_timing.so DAAM_NO_PAIRED_FINALIZE=1 DAAM_NO_PARTIAL_FINALIZE=1 timeout 120 python tools/fin_timing.py > gpurun_out/s8/fin_timing_unpaired.txt 2>&1
tail -3 gpurun_out/s8/pytest.log; grep -h finalize_us gpurun_out/s8/sweep.txt | cut -c1-150; tail -12 gpurun_out/s8/fin_timing_unpaired.txt

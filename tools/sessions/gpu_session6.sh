#!/bin/bash
mkdir -p gpurun_out/s6
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
for c in default 8 10 16 20 26 39 52; do
  if [ $c = default ]; then timeout 100 python tools/fin_sweep.py; else DAAM_FIN_CHUNKS=$c timeout 100 python tools/fin_sweep.py; fi
done > gpurun_out/s6/sweep_chunks.txt 2>&1
for l in k3w3 k4w3 k3w4; do
  for c in default 20 26; do
    if [ $c = default ]; then DAAM_HIP_LIB=$R/build/libdaam_fin_$l.so timeout 100 python tools/fin_sweep.py; else DAAM_FIN_CHUNKS=$c DAAM_HIP_LIB=$R/build/libdaam_fin_$l.so timeout 100 python tools/fin_sweep.py; fi
  done
done > gpurun_out/s6/sweep_libs.txt 2>&1
grep finalize_us gpurun_out/s6/*.txt

#!/bin/bash
# run tools/exp/pipe_hw.py under each tools/exp/libdaam_<tag>.so, compare the maps with the first tag's:  run_variants.sh <session> <tag>...
S=$1; shift
mkdir -p gpurun_out
for t in "$@"; do
  PIPE_HW_TAG=$t DAAM_HIP_LIB=$PWD/tools/exp/libdaam_$t.so timeout 200 python tools/exp/pipe_hw.py > gpurun_out/${S}_$t.log 2>&1
  echo "exit $?" >> gpurun_out/${S}_$t.log
done
python - "$@" <<'PY'
import sys, numpy as np
tags = sys.argv[1:]
for k in ('all', 'x2'):
    a = np.load(f'gpurun_out/fin_{tags[0]}_{k}.npy')
    for t in tags[1:]:
        b = np.load(f'gpurun_out/fin_{t}_{k}.npy')
        print(k, t, 'max abs diff', float(np.abs(a - b).max()), 'max', float(np.abs(a).max()))
PY

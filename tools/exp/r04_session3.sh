#!/bin/bash
# Round 4, GPU session 3: issue priority for the long-chain workgroups of the SD-v1.5 launch (-DDAAM_CHUNK_PRIO=1)
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_s3
mkdir -p "$out"
DAAM_HIP_LIB=tools/exp/libdaam_ctime_prio.so timeout 120 python tools/exp/chunk_timeline.py sd15 2> "$out/timeline_prio.log" > /dev/null
cat "$out/timeline_prio.log" | tail -5
cp gpurun_out/chunk_timeline_sd15.json "$out/chunk_timeline_sd15_prio.json"
DAAM_HIP_LIB=tools/exp/libdaam_prio.so timeout 120 python -m pytest tests/test_gpu_chunked.py -q -x 2>&1 | tail -1
timeout 90 python tools/exp/chunk_ab.py sd15 2> /dev/null > "$out/ab_default.json"; cat "$out/ab_default.json"; echo
DAAM_HIP_LIB=tools/exp/libdaam_prio.so timeout 90 python tools/exp/chunk_ab.py sd15 2> /dev/null > "$out/ab_prio.json"; cat "$out/ab_prio.json"; echo

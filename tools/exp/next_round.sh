#!/bin/bash
# First GPU session of the next round: what the last session of round 3 prepared without GPU time left.
# Run as ONE gpurun call (~2-3 min of box time):   gpurun --timeout 400 -- 'bash tools/exp/next_round.sh'
# Everything lands in gpurun_out/next_round/.
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/next_round
mkdir -p "$out"

# 1. the bf16 instantiation of tap_chunk_kernel (opt-in so far)
DAAM_TEST_UNVALIDATED=1 timeout 120 python -m pytest tests/test_gpu_chunked.py -q -x > "$out/chunked_tests.txt" 2>&1
tail -3 "$out/chunked_tests.txt"

# 2. does the footprint penalty of the SDXL tap depend on how Q / K are spread over allocator segments?
timeout 60 python tools/exp/pool_sweep.py 12 50 2> /dev/null > "$out/pool_default.json"
timeout 60 python tools/exp/pool_sweep.py --arena 12 50 2> /dev/null > "$out/pool_arena.json"
cat "$out/pool_default.json" "$out/pool_arena.json"

# 3. the chunked kernel with the next sub-step's DMAs ahead of the MFMAs
python - <<'PY'
from daam_amd import build
build.build_variant('tools/exp/libdaam_early.so', ['-DDAAM_CHUNK_EARLY_DMA=1'], verbose=False)
PY
DAAM_HIP_LIB=tools/exp/libdaam_early.so timeout 120 python -m pytest tests/test_gpu_chunked.py -q -x > "$out/chunked_tests_early.txt" 2>&1
tail -3 "$out/chunked_tests_early.txt"
timeout 60 python tools/exp/chunk_ab.py sd15 2> /dev/null > "$out/chunk_ab_default.json"
DAAM_HIP_LIB=tools/exp/libdaam_early.so timeout 60 python tools/exp/chunk_ab.py sd15 2> /dev/null > "$out/chunk_ab_early.json"
cat "$out/chunk_ab_default.json" "$out/chunk_ab_early.json"

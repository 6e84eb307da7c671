#!/bin/bash
# Round 4, first GPU session (one gpurun call): the evidence holes of the round-3 verdict + the footprint question.
#   gpurun --timeout 900 -- 'bash tools/exp/r04_session1.sh'
# tools/exp/libdaam_early.so (-DDAAM_CHUNK_EARLY_DMA=1) is built in the container beforehand and travels with the snapshot.
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_s1
mkdir -p "$out"
T0=$SECONDS
say() { echo "[s1 $((SECONDS - T0))s] $*"; }

# 1. bf16 instantiation of tap_chunk_kernel (never run so far) + the fp16 cases
DAAM_TEST_UNVALIDATED=1 timeout 200 python -m pytest tests/test_gpu_chunked.py -q -x > "$out/chunked_tests.txt" 2>&1
say "chunked (bf16 included): $(tail -1 "$out/chunked_tests.txt")"

# 2. the stack-level parity tests (headline configuration now at its own 50 steps) -> gpurun_out/fullsize_parity.json
timeout 400 python -m pytest tests/test_gpu_integration.py -q -x > "$out/integration_tests.txt" 2>&1
say "integration: $(tail -1 "$out/integration_tests.txt")"

# 3. footprint or re-use?
timeout 120 python tools/exp/footprint.py 2> "$out/footprint.log" > /dev/null
timeout 120 python tools/exp/footprint.py --arena 2> "$out/footprint_arena.log" > /dev/null
say "footprint:"; cat "$out/footprint.log" "$out/footprint_arena.log" | grep case

# 4. chunked kernel with the next sub-step's DMAs ahead of the MFMAs
if [ -f tools/exp/libdaam_early.so ]; then
  DAAM_HIP_LIB=tools/exp/libdaam_early.so timeout 200 python -m pytest tests/test_gpu_chunked.py -q -x > "$out/chunked_tests_early.txt" 2>&1
  say "chunked, early DMA: $(tail -1 "$out/chunked_tests_early.txt")"
  timeout 90 python tools/exp/chunk_ab.py sd15 2> /dev/null > "$out/chunk_ab_default.json"
  DAAM_HIP_LIB=tools/exp/libdaam_early.so timeout 90 python tools/exp/chunk_ab.py sd15 2> /dev/null > "$out/chunk_ab_early.json"
  say "chunk A/B default / early:"; cat "$out/chunk_ab_default.json"; echo; cat "$out/chunk_ab_early.json"; echo
fi

# 5. SD-v1.5 on the one-kernel launch: kernel stats + PMC passes -> gpurun_out/profiles_r04/
timeout 400 bash tools/profile_round.sh r04 sd15 50 50 30 5 > "$out/profile_sd15.log" 2>&1
say "profile sd15: $(tail -2 "$out/profile_sd15.log" | cut -c1-300)"
ls gpurun_out/profiles_r04 2> /dev/null
say done

#!/bin/bash
# round 6, GPU call 11: the -m gpu suite three times with four test processes sharing the GPU (timing-dependent failures show under contention;
# the multi-rank bench tests are left out: their throughput bound assumes the GPU to themselves)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 1200 python -m pytest tests/ -q -m gpu -n 4 --deselect tests/test_gpu_bench_multirank.py > gpurun_out/r6_contention_$i.txt 2>&1
  echo "round $i: $(tail -1 gpurun_out/r6_contention_$i.txt)"; grep -E "^FAILED|^ERROR" gpurun_out/r6_contention_$i.txt | head
done
ls gpurun_out/expand_mismatch_* 2>/dev/null

#!/bin/bash
# round 6, GPU call 7: what the driver runs at round end, in its order and form -- the -m gpu suite serially with -x, smoke, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
T0=$SECONDS
timeout 1500 python -m pytest tests/ -x -q -m gpu > gpurun_out/r06b_gpu_tests.txt 2>&1; echo "pytest rc=$? $((SECONDS - T0)) s: $(tail -1 gpurun_out/r06b_gpu_tests.txt)"
T0=$SECONDS
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06b_smoke.txt 2>&1; echo "smoke rc=$? $((SECONDS - T0)) s: $(tail -1 gpurun_out/r06b_smoke.txt)"
T0=$SECONDS
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06b_bench_line.json 2> gpurun_out/r06b_bench_err.txt; echo "bench rc=$? $((SECONDS - T0)) s, line $(wc -c < gpurun_out/r06b_bench_line.json) B"
cp gpurun_out/bench_full.json gpurun_out/r06b_bench_full.json
cat gpurun_out/r06b_bench_line.json

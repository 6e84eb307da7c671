#!/bin/bash
mkdir -p gpurun_out/s3
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s3/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s3/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-baselines > gpurun_out/s3/bench_asm.json 2> gpurun_out/s3/bench_asm.err
DAAM_HIP_LIB=$PWD/build/libdaam_fin_tied.so timeout 300 python bench.py --steps 20 --warmup 5 --no-baselines > gpurun_out/s3/bench_tied.json 2> gpurun_out/s3/bench_tied.err
grep -E "passed|failed" gpurun_out/s3/pytest.log | tail -3
python -c "
import json
for n in ('asm','tied'):
    d=json.load(open('gpurun_out/s3/bench_%s.json'%n)); print(n, d['roofline_finalize']['ms_per_launch'], d['roofline']['ms_per_launch'], d['value'])
"

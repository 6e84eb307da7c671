#!/bin/bash
mkdir -p gpurun_out/s10
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s10/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s10/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/s10/bench.json 2> gpurun_out/s10/bench.err
timeout 300 python bench.py --steps 20 --warmup 5 --workload sd15 --no-baselines > gpurun_out/s10/bench_sd15.json 2> gpurun_out/s10/bench_sd15.err
timeout 300 python bench.py --steps 5 --warmup 2 --workload sdxl2048 --denoise-steps 100 --no-baselines > gpurun_out/s10/bench_sdxl2048.json 2> gpurun_out/s10/bench_sdxl2048.err
grep -E "passed|failed|FAILED" gpurun_out/s10/pytest.log | tail -8
python -c "
import json
for n in ('bench','bench_sd15','bench_sdxl2048'):
    try:
        d=json.load(open('gpurun_out/s10/%s.json'%n)); print(n, d['value'], d['ms_per_step'], 'tap', d['roofline']['ms_per_launch'], d['roofline']['frac'], 'fin', d['roofline_finalize']['ms_per_launch'], d['roofline_finalize']['frac'], d.get('integrated'))
    except Exception as e: print(n, 'ERR', e)
"

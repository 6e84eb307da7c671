#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/s16
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > gpurun_out/s16/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s16/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s16/smoke.log 2>&1
/usr/bin/time -v timeout 900 python bench.py > gpurun_out/s16/bench_sdxl1024.json 2> gpurun_out/s16/bench_sdxl1024.err
timeout 300 python bench.py --steps 20 --warmup 5 --workload sd15 --no-baselines > gpurun_out/s16/bench_sd15.json 2> gpurun_out/s16/bench_sd15.err
timeout 300 python bench.py --steps 5 --warmup 2 --workload sdxl2048 --denoise-steps 100 --no-baselines > gpurun_out/s16/bench_sdxl2048.json 2> gpurun_out/s16/bench_sdxl2048.err
tools/ubench_issue > gpurun_out/s16/ubench_issue.txt 2>&1
tools/ubench_fin > gpurun_out/s16/ubench_fin.txt 2>&1
bash tools/profile_round.sh r02 sdxl1024 50 50 30 5 > gpurun_out/s16/prof_sdxl1024.log 2>&1
bash tools/profile_round.sh r02 sd15 50 50 30 5 > gpurun_out/s16/prof_sd15.log 2>&1
bash tools/profile_round.sh r02 sdxl2048 100 24 4 2 > gpurun_out/s16/prof_sdxl2048.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/s16/pytest.log | tail -5; tail -2 gpurun_out/s16/smoke.log; grep -E "Elapsed|Maximum resident" gpurun_out/s16/bench_sdxl1024.err
python -c "
import json
for n in ('bench_sdxl1024','bench_sd15','bench_sdxl2048'):
    try:
        d=json.load(open('gpurun_out/s16/%s.json'%n)); print(n, d['value'], d['ms_per_step'], 'tap', d['roofline']['ms_per_launch'], d['roofline']['frac'], 'fin', d['roofline_finalize']['ms_per_launch'], d['roofline_finalize']['frac'], d.get('integrated',{}) and d['integrated'].get('overhead_ms_per_step'))
    except Exception as e: print(n, 'ERR', e)
"

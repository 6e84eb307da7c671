#!/bin/bash
# full round check: GPU tests, smoke, profiles of the three workloads (then copied to profiles/ so that the bench lines of
# the same session can quote them), the three bench lines, the attend micro-benchmark with its kernel trace
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/s27; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --timeout 900 > $O/pytest.log 2>&1
echo "pytest rc=$?" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
bash tools/profile_round.sh r02 sdxl1024 50 50 30 5 > $O/prof_sdxl1024.log 2>&1
bash tools/profile_round.sh r02 sd15 50 50 30 5 > $O/prof_sd15.log 2>&1
bash tools/profile_round.sh r02 sdxl2048 100 24 4 2 > $O/prof_sdxl2048.log 2>&1
cp gpurun_out/profiles_r02/r02_counters.json gpurun_out/profiles_r02/hbm_traffic.json profiles/
T0=$(date +%s)
timeout 900 python bench.py > $O/bench_sdxl1024.json 2> $O/bench_sdxl1024.err
T1=$(date +%s)
echo "default bench wall seconds: $((T1 - T0))" > $O/bench_wall.txt
timeout 300 python bench.py --steps 20 --warmup 5 --workload sd15 --no-baselines > $O/bench_sd15.json 2> $O/bench_sd15.err
timeout 300 python bench.py --steps 5 --warmup 2 --workload sdxl2048 --denoise-steps 100 --no-baselines > $O/bench_sdxl2048.json 2> $O/bench_sdxl2048.err
timeout 300 python tools/attend_bench.py 50 5 > $O/attend_bench.json 2> $O/attend_bench.err
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_attend -- python $R/tools/attend_bench.py 10 2 > $R/$O/prof_attend.log 2>&1 )
f=$(find $O/prof_attend -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" > $O/attend_kernel_stats.csv
rm -rf $O/prof_attend
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -5; tail -2 $O/smoke.log; cat $O/bench_wall.txt
python -c "
import json
for n in ('bench_sdxl1024','bench_sd15','bench_sdxl2048'):
    try:
        d=json.load(open('$O/%s.json'%n)); print(n, d['value'], d['ms_per_step'], 'tap', d['roofline']['ms_per_launch'], d['roofline']['frac'], (d.get('roofline_issue') or {}).get('frac'), 'fin', d['roofline_finalize']['ms_per_launch'], d['roofline_finalize']['frac'], (d.get('roofline_finalize_issue') or {}).get('frac'), (d.get('integrated') or {}).get('overhead_ms_per_step'))
    except Exception as e: print(n, 'ERR', e)
"
cat $O/attend_bench.json

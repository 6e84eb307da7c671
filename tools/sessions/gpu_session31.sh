#!/bin/bash
# final evidence of the round on the final build: profiles of the three workloads (copied to profiles/ so that the bench
# lines of the same session quote them), bench lines (the default one three times: run-to-run spread), attend bench + its
# kernel trace + FETCH_SIZE / WRITE_SIZE of the attend kernel with and without the fused tap
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/s31; mkdir -p $O
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
for i in 2 3; do timeout 300 python bench.py --no-baselines --no-integrated > $O/bench_sdxl1024_run$i.json 2> /dev/null; done
timeout 300 python bench.py --steps 20 --warmup 5 --workload sd15 --no-baselines > $O/bench_sd15.json 2> $O/bench_sd15.err
timeout 300 python bench.py --steps 5 --warmup 2 --workload sdxl2048 --denoise-steps 100 --no-baselines > $O/bench_sdxl2048.json 2> $O/bench_sdxl2048.err
timeout 300 python tools/attend_bench.py 50 5 > $O/attend_bench.json 2> $O/attend_bench.err
( cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_attend -- python $R/tools/attend_bench.py 10 2 > $R/$O/prof_attend.log 2>&1
  for which in attend attend_fused_tap; do for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/pmc_${which}_$c -- python $R/tools/attend_bench.py 4 1 $which > $R/$O/pmc_${which}_$c.log 2>&1
  done; done )
f=$(find $O/prof_attend -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -12 "$f" > $O/attend_kernel_stats.csv
python - <<'PY' > $O/attend_pmc.json
import csv, glob, json, statistics
out = {}
for which in ('attend', 'attend_fused_tap'):
    rec = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        fs = glob.glob(f'gpurun_out/s31/pmc_{which}_{c}/**/*counter_collection.csv', recursive=True)
        vals = [float(r['Counter_Value']) for f in fs for r in csv.DictReader(open(f)) if 'attend_kernel' in r['Kernel_Name'] and r['Counter_Name'] == c]
        if vals:
            rec[c + '_KiB_mean_per_call'] = round(statistics.fmean(vals), 1)
            rec['calls'] = len(vals)
    if 'FETCH_SIZE_KiB_mean_per_call' in rec and 'WRITE_SIZE_KiB_mean_per_call' in rec:
        rec['hbm_bytes_mean_per_call'] = int((2 * rec['FETCH_SIZE_KiB_mean_per_call'] + rec['WRITE_SIZE_KiB_mean_per_call']) * 1024)
    out[which] = rec
out['method'] = 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) of tools/attend_bench.py 4 1 <loop>; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950 correction as for the other kernels); mean over the 60 SDXL layer calls of a step'
print(json.dumps(out, indent=1))
PY
rm -rf $O/prof_attend $O/pmc_*_FETCH_SIZE $O/pmc_*_WRITE_SIZE
grep -E "passed|failed|FAILED|rc=" $O/pytest.log | tail -4; tail -1 $O/smoke.log; cat $O/bench_wall.txt
python -c "
import json
for n in ('bench_sdxl1024','bench_sdxl1024_run2','bench_sdxl1024_run3','bench_sd15','bench_sdxl2048'):
    try:
        d=json.load(open('$O/%s.json'%n)); print(n, d['value'], d['ms_per_step'], 'tap', d['roofline']['ms_per_launch'], d['roofline']['frac'], (d.get('roofline_issue') or {}).get('frac'), 'fin', d['roofline_finalize']['ms_per_launch'], d['roofline_finalize']['frac'], (d.get('roofline_finalize_issue') or {}).get('frac'), (d.get('integrated') or {}).get('overhead_ms_per_step'))
    except Exception as e: print(n, 'ERR', e)
"
cat $O/attend_bench.json; cat $O/attend_pmc.json

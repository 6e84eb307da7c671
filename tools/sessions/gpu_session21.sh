#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/s21; mkdir -p $O
timeout 300 python tools/attend_bench.py 50 5 > $O/attend_bench.json 2> $O/attend_bench.err
cat $O/attend_bench.json; tail -n 3 $O/attend_bench.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/tools/attend_bench.py 10 2 > $O/prof.log 2>&1
cd $R
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print(r['Name'][:100], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
cp "$f" $O/attend_kernel_stats.csv
rm -rf $O/prof

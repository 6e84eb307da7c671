#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/s20; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
PROBE_PHASES=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/tools/overhead_probe.py 10 3 > $O/probe_prof.log 2>&1
cd $R
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:25]:
    print(r['Name'][:90], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'])
PY
cp "$f" $O/kernel_stats.csv
rm -rf $O/prof

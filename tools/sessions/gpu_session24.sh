#!/bin/bash
# vector-memory pipeline counters of the tap launch (is the fragment-shaped Q fetch a TA / TCP problem?)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/s24; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -oE "\b(TA|TCP|TD|TCC)_[A-Z0-9_a-z]+" $O/avail.txt | sort -u > $O/avail_names.txt
wc -l $O/avail_names.txt
PM="python $R/bench.py --no-baselines --no-integrated --steps 3 --warmup 1"
pass() { # name counters...
  n=$1; shift
  have=""
  for c in "$@"; do grep -qx "$c" $O/avail_names.txt && have="$have $c"; done
  [ -z "$have" ] && { echo "$n: none available"; return; }
  timeout 300 rocprofv3 --kernel-trace --pmc $have --output-format csv -d $O/$n -- $PM > $O/$n.log 2>&1
  f=$(find $O/$n -name '*counter_collection.csv' | head -1)
  python - "$f" <<'PY'
import csv, sys, collections, statistics
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'tap_d64' in r['Kernel_Name'] or 'finalize_up32' in r['Kernel_Name']:
        acc[(r['Kernel_Name'][:40], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(k, 'n', len(v), 'median', statistics.median(v))
PY
  rm -rf $O/$n
}
pass ta TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum
pass tcp2 TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass td TD_TD_BUSY_sum TD_BUSY_avr TD_TC_STALL_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum

#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/s26; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_attend.py -m gpu -q -x --timeout 600 -k "tap or fused or golden or attend" > $O/pytest.log 2>&1
tail -n 4 $O/pytest.log
for i in 1 2; do
timeout 300 python bench.py --no-baselines --no-integrated > $O/bench_$i.json 2> $O/bench_$i.err
done
python -c "
import json
for n in ('bench_1','bench_2'):
    try:
        d=json.load(open('$O/%s.json'%n)); print(n, d['value'], d['ms_per_step'], 'tap', d['roofline']['ms_per_launch'], d['roofline']['frac'], (d.get('roofline_issue') or {}).get('clock',{}).get('mhz_median_under_load'), 'fin', d['roofline_finalize']['ms_per_launch'])
    except Exception as e: print(n, 'ERR', e)
"
cd /tmp; export TMPDIR=/tmp
PM="python $R/bench.py --no-baselines --no-integrated --steps 3 --warmup 1"
pass() {
  n=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- $PM > $O/$n.log 2>&1
  f=$(find $O/$n -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections, statistics
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'tap_d64' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print('tap_d64_kernel', k, 'launches', len(v), 'median per launch', statistics.median(v))
PY
  rm -rf $O/$n
}
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum > $O/tcp_after.txt
pass td TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TD_TC_STALL_sum >> $O/tcp_after.txt
cat $O/tcp_after.txt

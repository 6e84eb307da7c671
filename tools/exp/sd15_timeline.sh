#!/bin/bash
# kernel timeline of SD-v1.5 tap launches: start / end of the three kernels of each flush relative to the first start
#   sd15_timeline.sh <tag> [lib.so]
set -u
TAG=$1; LIB=${2:-}
R=$(pwd); O=$R/gpurun_out/tl_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
[ -n "$LIB" ] && export DAAM_HIP_LIB=$LIB
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/bench.py --no-baselines --no-integrated --no-other-configs --workload sd15 --steps 6 --warmup 2 > $O/log.txt 2>&1
cd $R
python - $O <<'PY'
import sys, glob, csv, collections
O = sys.argv[1]
f = glob.glob(O + '/tr/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'tap_' in r['Kernel_Name'] or 'finalize' in r['Kernel_Name'] or 'upload' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# group into flushes: a new group when a tap kernel starts > 50 us after the previous group's last end
groups, cur, last_end = [], [], None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if 'tap_' in r['Kernel_Name'] and (last_end is None or s - last_end > 20000) and cur:
        groups.append(cur); cur = []
    cur.append(r); last_end = max(last_end or 0, e)
groups.append(cur)
def short(n):
    for k in ('tap_wide_kernelIDF16_Lb1ELi5', 'tap_wide_kernelIDF16_Lb1ELi3', 'tap_d64', 'finalize_up32_pipe', 'finalize_up_kernel', 'upload'):
        if k in n: return {'tap_wide_kernelIDF16_Lb1ELi5': 'wide5', 'tap_wide_kernelIDF16_Lb1ELi3': 'wide3'}.get(k, k)
    return n[:30]
out = open(O + '/timeline.txt', 'w')
taps = [r for r in rows if 'tap_' in r['Kernel_Name']]
# flushes = runs of tap kernels whose starts are within 100 us of the run's first start
flushes, cur = [], []
for r in taps:
    s = int(r['Start_Timestamp'])
    if cur and s - int(cur[0]['Start_Timestamp']) > 100000 and len({short(x['Kernel_Name']) for x in cur}) >= 3:
        flushes.append(cur); cur = []
    cur.append(r)
flushes.append(cur)
for g in flushes[3::max(1, len(flushes) // 14)][:16]:
    t0 = min(int(r['Start_Timestamp']) for r in g)
    end = max(int(r['End_Timestamp']) for r in g)
    line = f'flush {(end - t0) / 1e3:4.0f} us: ' + '  '.join(f"{short(r['Kernel_Name'])} {(int(r['Start_Timestamp']) - t0) / 1e3:.0f}-{(int(r['End_Timestamp']) - t0) / 1e3:.0f}" for r in g)
    print(line); out.write(line + '\n')
PY
rm -rf $O/tr

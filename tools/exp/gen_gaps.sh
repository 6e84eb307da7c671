#!/bin/bash
# Where a generation's wall time goes on the GPU: rocprofv3 --kernel-trace of the headline bench leg, then the gaps between the kernels
# of consecutive generations (upload -> tap -> finalize -> next upload).   gpurun -- 'bash tools/exp/gen_gaps.sh'
set -u
R=$(cd "$(dirname "$0")/../.." && pwd)
O=$R/gpurun_out/gen_gaps; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/bench.py --no-baselines --no-integrated --no-pmc --no-other-configs --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.log
python - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
f = glob.glob(O + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
def short(n):
    for k in ('tap_d64_kernel', 'finalize_up32_pipe_kernel', 'upload_kernel', 'clock_monitor'):
        if k in n: return k
    return n[:30]
seq = [(s, e, short(n)) for s, e, n in ev]
# steady state: the last 30 generations of (upload, tap, finalize) triples
taps = [i for i, x in enumerate(seq) if x[2] == 'tap_d64_kernel']
out = []
for a, b in zip(taps[:-1], taps[1:]):
    seg = seq[a:b + 1]
    names = [x[2] for x in seg]
    if names.count('finalize_up32_pipe_kernel') != 1: continue
    t_end = seg[0][1]
    fin = next(x for x in seg if x[2] == 'finalize_up32_pipe_kernel')
    ups = [x for x in seg[1:] if x[2] == 'upload_kernel']
    nxt = seg[-1]
    out.append(dict(tap_us=(seg[0][1] - seg[0][0]) / 1e3, tap_to_fin_us=(fin[0] - t_end) / 1e3, fin_us=(fin[1] - fin[0]) / 1e3,
                    fin_to_next_tap_us=(nxt[0] - fin[1]) / 1e3, uploads=len(ups), upload_us=sum(u[1] - u[0] for u in ups) / 1e3,
                    others=[n for n in names[1:-1] if n not in ('finalize_up32_pipe_kernel', 'upload_kernel')], period_us=(nxt[0] - seg[0][0]) / 1e3))
import statistics as st
tail = out[-30:]
summ = {k: round(st.median(x[k] for x in tail), 2) for k in ('tap_us', 'tap_to_fin_us', 'fin_us', 'fin_to_next_tap_us', 'upload_us', 'period_us')}
summ['uploads'] = tail[-1]['uploads']; summ['others'] = tail[-1]['others']; summ['n'] = len(tail)
print(json.dumps(summ))
json.dump(dict(summary=summ, generations=tail), open(O + '/gaps.json', 'w'), indent=1)
PY
tail -1 $O/bench.json | cut -c1-300
rm -rf $O/trace

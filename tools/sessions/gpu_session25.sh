#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/s25; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_attend.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1
tail -n 8 $O/pytest.log
for i in 1 2; do
timeout 300 python bench.py --no-baselines --no-integrated > $O/bench_$i.json 2> $O/bench_$i.err
done
timeout 300 python bench.py --no-baselines --no-integrated --workload sd15 --steps 20 --warmup 5 > $O/bench_sd15.json 2> $O/bench_sd15.err
python -c "
import json
for n in ('bench_1','bench_2','bench_sd15'):
    try:
        d=json.load(open('$O/%s.json'%n)); print(n, d['value'], d['ms_per_step'], 'tap', d['roofline']['ms_per_launch'], d['roofline']['frac'], (d.get('roofline_issue') or {}).get('clock',{}).get('mhz_median_under_load'), 'fin', d['roofline_finalize']['ms_per_launch'])
    except Exception as e: print(n, 'ERR', e)
"

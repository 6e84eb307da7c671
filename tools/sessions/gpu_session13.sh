#!/bin/bash
mkdir -p gpurun_out/s13
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
for v in abl1 abl2; do
DAAM_HIP_LIB=$R/build/libdaam_tap_$v.so timeout 300 python bench.py --steps 10 --warmup 3 --no-baselines --no-integrated > gpurun_out/s13/bench_$v.json 2> gpurun_out/s13/bench_$v.err
done
timeout 300 python bench.py --steps 10 --warmup 3 --no-baselines --no-integrated > gpurun_out/s13/bench.json 2> gpurun_out/s13/bench.err
python -c "
import json
for n in ('bench','bench_abl1','bench_abl2'):
    try:
        d=json.load(open('gpurun_out/s13/%s.json'%n)); print(n, d['value'], 'tap', d['roofline']['ms_per_launch'], 'clock', d['roofline_issue']['clock'])
    except Exception as e: print(n, 'ERR', e)
"

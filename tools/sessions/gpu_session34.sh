#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/s34; mkdir -p $O
timeout 900 python bench.py > $O/bench_sdxl1024.json 2> $O/bench_sdxl1024.err
python -c "
import json
d=json.load(open('$O/bench_sdxl1024.json')); print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_issue'].get('frac'), d['integrated'])"

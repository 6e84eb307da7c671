#!/bin/bash
# the other two workload lines on the final commit + the driver's launcher form at N = 1
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=gpurun_out/s36; mkdir -p $O
timeout 300 python bench.py --steps 20 --warmup 5 --workload sd15 --no-baselines > $O/bench_sd15.json 2> $O/bench_sd15.err
timeout 300 python bench.py --steps 5 --warmup 2 --workload sdxl2048 --denoise-steps 100 --no-baselines > $O/bench_sdxl2048.json 2> $O/bench_sdxl2048.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 5 --warmup 2 --no-baselines --no-integrated > $O/bench_launcher.json 2> $O/bench_launcher.err
python -c "
import json
for n in ('bench_sd15','bench_sdxl2048','bench_launcher'):
    try:
        d=json.load(open('$O/%s.json'%n)); print(n, d['value'], d['n_gpus'], d['ms_per_step'], 'tap', d['roofline']['ms_per_launch'], d['roofline']['frac'], (d.get('roofline_issue') or {}).get('frac'), 'fin', d['roofline_finalize']['ms_per_launch'], d['roofline_finalize']['frac'])
    except Exception as e: print(n, 'ERR', e)
"
tail -n 2 $O/bench_launcher.err

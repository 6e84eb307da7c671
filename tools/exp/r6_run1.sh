#!/bin/bash
# round 6, GPU call 2: parity of the new / changed paths, then A/B on ONE box:
#  (a) headline: the in-tree library against tools/exp/libdaam_A.so (commit e3a62d0: before head_minor / pow2 / gate changes)
#  (b) SD-v1.5: DAAM_SLAB_ORDER x DAAM_SLAB_TAIL
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_slab.py tests/test_gpu_bench_multirank.py tests/test_gpu_parity.py -q -m gpu -k "slab or eight_wave or n_rows or bench or prepare or wide_logit" 2>&1 | tail -6
for i in 1 2 3 4 5; do timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "test_trace_api_matches_reference_golden and sdxl_f32" 2>&1 | tail -1; done
python - <<'PY'
import json, os, subprocess, sys
def bench(args, env_extra):
    env = dict(os.environ, BENCH_FULL_RECORD='/tmp/bench_full_ab.json', **env_extra)
    for k, v in env_extra.items():
        if v is None: env.pop(k)
    p = subprocess.run([sys.executable, 'bench.py', '--no-baselines', '--no-integrated', '--no-pmc', '--no-other-configs', *args], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=400)
    r = json.load(open('/tmp/bench_full_ab.json'))
    return dict(maps_per_s=r['value'], tap_ms=r['roofline']['ms_per_launch'], tap_ms_iso=r['roofline']['ms_per_launch_isolated'],
                sustained=r.get('sustained_maps_per_s'), sustained_tap_ms=r.get('sustained_tap_ms'), fin_us=round(r['roofline_finalize']['ms_per_launch'] * 1e3, 1),
                kernel=r['roofline']['kernel'][:20])
rows = []
for i in range(3):
    for tag, env in (('A', dict(DAAM_HIP_LIB='tools/exp/libdaam_A.so')), ('new', {})):
        row = dict(leg='headline', lib=tag, **bench(['--steps', '30', '--warmup', '5'], env)); rows.append(row); print(row, flush=True)
for i in range(2):
    for order, tail in ((0, 25), (1, 25), (1, 18), (1, 12), (1, 0), (1, 31), (0, 0)):
        row = dict(leg='sd15', order=order, tail=tail, **bench(['--workload', 'sd15', '--steps', '100', '--warmup', '10', '--no-sustained'],
                                                                dict(DAAM_SLAB_ORDER=str(order), DAAM_SLAB_TAIL=str(tail)))); rows.append(row); print(row, flush=True)
json.dump(rows, open('gpurun_out/r6_run1_ab.json', 'w'), indent=1)
PY

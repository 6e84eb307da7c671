#!/bin/bash
# round 6, GPU call 9: workgroup numbering of the head_dim-64 tap launch -- groups of 2 / 4 adjacent heads tile-major, chunks of 70 workgroups dealt
# round-robin to the XCDs -- against the shipped (head, tile) numbering with contiguous XCD ranges; variant libraries built by
# tools/exp/build_variant.sh from tools/exp/patches/tap_d64_numbering_*.patch.  Parity first (the sums must not change), then the headline leg.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in pair quad xcd70; do
  DAAM_HIP_LIB=tools/exp/libdaam_$v.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "eight_wave or tap_qk_vs_oracle or golden" 2>&1 | tail -1
done
python - <<'PY'
import json, os, subprocess, sys
def bench(args, env_extra):
    env = dict(os.environ, BENCH_FULL_RECORD='/tmp/bench_full_ab.json', **env_extra)
    p = subprocess.run([sys.executable, 'bench.py', '--no-baselines', '--no-integrated', '--no-pmc', '--no-other-configs', *args], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    if p.returncode: print(p.stderr[-2000:])
    r = json.load(open('/tmp/bench_full_ab.json'))
    return dict(maps_per_s=r['value'], tap_ms=r['roofline']['ms_per_launch'], tap_ms_iso=r['roofline']['ms_per_launch_isolated'],
                sustained=r.get('sustained_maps_per_s'), sustained_tap_ms=r.get('sustained_tap_ms'))
rows = []
for i in range(3):
    for tag in ('default', 'pair', 'quad', 'xcd70'):
        env = {} if tag == 'default' else dict(DAAM_HIP_LIB=f'tools/exp/libdaam_{tag}.so')
        row = dict(lib=tag, **bench(['--steps', '30', '--warmup', '5'], env)); rows.append(row); print(row, flush=True)
json.dump(rows, open('gpurun_out/r6_run9_ab.json', 'w'), indent=1)
PY

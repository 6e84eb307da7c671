#!/bin/bash
# round 6, GPU call 5: pipelined finalize for bf16 (three pass-1 MFMAs) / f32 sums -- parity, bench legs, kernel stats
cd "$GRAFT_REPO_ROOT" || exit 1
R=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -n 4 -k "finalize" 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_integration.py -q -m gpu -x -k "full_size_parity" 2>&1 | tail -25
cat gpurun_out/fullsize_parity.json
python - <<'PY'
import json, os, subprocess, sys
def bench(args, env_extra):
    env = dict(os.environ, BENCH_FULL_RECORD='/tmp/bench_full_ab.json', **env_extra)
    p = subprocess.run([sys.executable, 'bench.py', '--no-baselines', '--no-integrated', '--no-pmc', '--no-other-configs', '--no-sustained', *args], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=400)
    if p.returncode: print(p.stderr[-2000:])
    r = json.load(open('/tmp/bench_full_ab.json'))
    return dict(maps_per_s=r['value'], tap_ms=r['roofline']['ms_per_launch'], fin_us=round(r['roofline_finalize']['ms_per_launch'] * 1e3, 1),
                fin_frac=r['roofline_finalize'].get('frac'), fin_kernel=r['roofline_finalize'].get('kernel', '')[:60])
rows = []
for i in range(2):
    for wl in ('sdxl1024', 'sdxl1024_bf16', 'sdxl1024_f32acc'):
        for tag, env in (('A', dict(DAAM_HIP_LIB='tools/exp/libdaam_A.so')), ('new', {})):
            row = dict(leg=wl, lib=tag, **bench(['--workload', wl, '--steps', '20', '--warmup', '5'], env)); rows.append(row); print(row, flush=True)
json.dump(rows, open('gpurun_out/r6_run5_ab.json', 'w'), indent=1)
PY
cd /tmp
for wl in sdxl1024 sdxl1024_bf16 sdxl1024_f32acc; do
  O=$R/gpurun_out/prof_r6_$wl; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-baselines --no-integrated --no-other-configs --no-pmc --no-sustained --workload $wl --steps 20 --warmup 2 > $O/stats.log 2>&1
  f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
  echo "== $wl"; head -8 "$f" | cut -c1-220
  cp "$f" $R/gpurun_out/r6_${wl}_kernel_stats.csv
  rm -rf $O/stats
done

#!/bin/bash
# Round 4, GPU session 2: the finalize-prepare path, bf16 chunk routing, SD-v1.5 launch timeline, bench line with the in-run PMC leg.
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_s2
mkdir -p "$out"
T0=$SECONDS
say() { echo "[s2 $((SECONDS - T0))s] $*"; }
timeout 300 python -m pytest tests/test_gpu_chunked.py tests/test_gpu_parity.py -q -x -k "chunked or finalize or views or normalize" > "$out/tests_a.txt" 2>&1
say "chunked + finalize tests: $(tail -1 "$out/tests_a.txt")"
grep -E "FAILED|Error|assert" "$out/tests_a.txt" | head -20
DAAM_HIP_LIB=tools/exp/libdaam_ctime.so timeout 120 python tools/exp/chunk_timeline.py sd15 2> "$out/timeline_sd15.log" > /dev/null
say "timeline sd15:"; cat "$out/timeline_sd15.log" | tail -8
timeout 400 python bench.py --no-baselines --no-integrated > "$out/bench.json" 2> "$out/bench.log"
say "bench: $(tail -3 "$out/bench.log")"
python - <<'PY'
import json
r = json.loads([l for l in open('gpurun_out/r04_s2/bench.json') if l.startswith('{')][0])
print('value', r['value'], 'ms_per_step', r['ms_per_step'])
print('roofline', {k: v for k, v in r['roofline'].items() if k not in ('traffic_per_kernel', 'traffic_source', 'kernel')})
print('per_kernel', r['roofline'].get('traffic_per_kernel'))
print('issue', r['roofline_issue'])
print('fin', {k: v for k, v in r['roofline_finalize'].items() if k != 'kernel'})
print('fin_issue', {k: v for k, v in r['roofline_finalize_issue'].items() if k not in ('kernel', 'model')})
for k, v in r['other_configs'].items():
    print(k, v['value'], {a: b for a, b in v['roofline'].items() if a in ('frac', 'ms_per_launch', 'traffic')}, {a: b for a, b in v['roofline_finalize'].items() if a in ('frac', 'ms_per_launch')})
PY
timeout 900 python -m pytest tests/test_gpu_bench_multirank.py tests/test_gpu_distributed.py -q -x > "$out/tests_b.txt" 2>&1
say "multirank tests: $(tail -1 "$out/tests_b.txt")"
grep -E "FAILED|Error" "$out/tests_b.txt" | head -20
say done

#!/bin/bash
# Round 4, GPU session 4: softmax without a reference point + pre-masked padding tokens: parity suites, then the bench line
set -u
cd "$(dirname "$0")/../.."
out=gpurun_out/r04_s4
mkdir -p "$out"
T0=$SECONDS
say() { echo "[s4 $((SECONDS - T0))s] $*"; }
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_attend.py tests/test_gpu_chunked.py tests/test_gpu_integration.py tests/test_gpu_fullsize.py -q -x > "$out/tests.txt" 2>&1
say "parity / attend / chunked / integration / fullsize: $(tail -1 "$out/tests.txt")"
grep -E "^FAILED|^ERROR|Error|assert " "$out/tests.txt" | head -20
for i in 1 2; do
timeout 300 python bench.py --no-baselines --no-integrated --no-pmc > "$out/bench$i.json" 2> "$out/bench$i.log"
python - "$out/bench$i.json" <<'PY'
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
print('value', r['value'], 'ms_per_step', r['ms_per_step'], 'tap', r['roofline']['ms_per_launch'], r['roofline']['frac'], 'clock', r['roofline_issue']['clock']['mhz_median_under_load'],
      'fin', r['roofline_finalize']['ms_per_launch'], r['roofline_finalize']['frac'])
for k, v in r['other_configs'].items():
    print(' ', k, v['value'], v['roofline']['ms_per_launch'], v['roofline']['frac'], 'fin', v['roofline_finalize']['ms_per_launch'])
PY
done
say done

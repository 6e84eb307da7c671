#!/bin/bash
# does the length of the untimed warm-up change what the driver's form of the bench measures?  (BENCH_MIN_WARM generations before the timed 20)
cd "$(dirname "$0")/../.."
for w in ${WARMS:-20 100 20 100 300}; do
  BENCH_MIN_WARM=$w python bench.py --gpus 1 --steps 20 --warmup 5 --no-pmc --no-baselines --no-integrated --no-other-configs 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('W', $w, r['value'], r['ms_per_step'], r['roofline']['ms_per_launch'], r['roofline_finalize']['ms_per_launch'])"
done

#!/bin/bash
# eight-wave / four-wave workgroups of the head_dim-64 tap (DAAM_TAP_W8=1 / 0), bench.py's headline leg alternating on one box
cd "$(dirname "$0")/../.."
for i in 1 2 3; do for w in 1 0; do
  DAAM_TAP_W8=$w python bench.py --no-baselines --no-integrated --no-pmc --no-other-configs --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('W8=$w', r['value'], r['roofline']['ms_per_launch'], r['roofline']['ms_per_launch_isolated'])"
done; done

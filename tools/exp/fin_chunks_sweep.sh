#!/bin/bash
# finalize of the SDXL-2048 workload (x0.5 + same-size classes, 883 MB) against DAAM_FIN_CHUNKS (key chunks per token of the streaming classes)
cd "$(dirname "$0")/../.."
for c in ${CHUNKS:-0 8 16 24 32 48 64}; do
  r=$(DAAM_FIN_CHUNKS=$c python bench.py --workload sdxl2048 --no-baselines --no-integrated --no-pmc --steps 4 --warmup 2 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['roofline_finalize']['ms_per_launch'], r['roofline_finalize']['frac'], r['value'])")
  echo "DAAM_FIN_CHUNKS=$c fin_ms frac maps/s: $r"
done

#!/bin/bash
# finalize of the SDXL-1024 workload against DAAM_FIN_PIPE_CHUNKS (key chunks per token of the pipelined x2 kernel; default 13)
cd "$(dirname "$0")/../.."
for c in ${CHUNKS:-0 9 11 13 15 18 22 26}; do
  r=$(DAAM_FIN_PIPE_CHUNKS=$c python bench.py --no-other-configs --no-baselines --no-integrated --no-pmc --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['roofline_finalize']['ms_per_launch'], r['roofline_finalize']['frac'], r['value'])")
  echo "DAAM_FIN_PIPE_CHUNKS=$c fin_ms frac maps/s: $r"
done

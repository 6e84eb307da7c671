#!/bin/bash
# finalize of the SD-v1.5 workload (x2 + same-size keys on the pipelined kernel, x4 keys on finalize_up_kernel<16>) against the two chunk counts
cd "$(dirname "$0")/../.."
run() { python bench.py --workload sd15 --no-baselines --no-integrated --no-pmc --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(r['roofline_finalize']['ms_per_launch'], r['value'])"; }
echo "default: $(run)"
for c in 1 2 3 4 6; do echo "DAAM_FIN_UP_CHUNKS=$c: $(DAAM_FIN_UP_CHUNKS=$c run)"; done
for c in 2 3 4 5 7 10; do echo "DAAM_FIN_PIPE_CHUNKS=$c: $(DAAM_FIN_PIPE_CHUNKS=$c run)"; done
echo "default: $(run)"

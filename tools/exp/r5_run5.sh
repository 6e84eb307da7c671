#!/bin/bash
# round 5, fifth GPU call: daam_attend with Q through LDS rows (default) against the register loads (-DDAAM_ATTEND_QLDS=0), parity of the attend paths
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2 3; do
  timeout 120 python tools/attend_bench.py 20 3 > gpurun_out/r5_5_attend_qlds_$rep.json 2>/dev/null
  DAAM_HIP_LIB=tools/exp/libdaam_noqlds.so timeout 120 python tools/attend_bench.py 20 3 > gpurun_out/r5_5_attend_regs_$rep.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5_5_attend_*.json')):
    try:
        r=json.loads([l for l in open(f) if l.startswith('{')][0])
        print(f, 'attend', r['attend']['ms_per_step'], 'fused', r['attend_fused_tap']['ms_per_step'], 'sdpa', r['torch_sdpa']['ms_per_step'], 'frac', r['attend']['frac_of_hbm_peak'], r['sums_bit_identical'])
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 900 python -m pytest tests/test_gpu_attend.py tests/test_gpu_processor.py tests/test_gpu_slab.py -x -q -m gpu 2>&1 | tail -4

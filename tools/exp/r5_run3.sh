#!/bin/bash
# round 5, third GPU call: Q2 (two-steps-ahead half Q tile) A/B on the headline, parity of the new forms, the new bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
A="--no-baselines --no-integrated --no-other-configs --no-pmc --no-sustained --steps 200 --warmup 10"
for rep in 1 2 3; do
  for q2 in 1 0; do
    DAAM_TAP_Q2=$q2 timeout 200 python bench.py $A > gpurun_out/r5_3_q2_${q2}_$rep.json 2>/dev/null
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5_3_q2_*.json')):
    try:
        r=json.load(open(f)); ro=r['roofline']
        print(f, r['value'], 'tap ms region', ro['ms_per_launch'], 'iso', ro['ms_per_launch_isolated'], 'clk', r['roofline_issue']['clock']['mhz_median_under_load'])
    except Exception as e:
        print(f, 'ERR', e)
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_integration.py tests/test_gpu_slab.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r5_3_bench_driver_form.json 2> gpurun_out/r5_3_bench_driver_form.err; tail -25 gpurun_out/r5_3_bench_driver_form.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r5_3_bench_driver_form.json'))
ro=r['roofline']
print('value', r['value'], 'tap', ro['ms_per_launch'], ro['frac'], 'traffic x', ro.get('traffic_over_algorithmic'), 'sustained', r.get('sustained_maps_per_s'), r.get('sustained_tap_ms'))
print('fin', r['roofline_finalize']['ms_per_launch'], r['roofline_finalize']['frac'], r['roofline_finalize'].get('frac_of_max_floor'), r['roofline_finalize'].get('traffic_measured_in_run'))
for k,o in r.get('other_configs',{}).items():
    print(k, o['value'], 'tap', o['roofline']['ms_per_launch'], o['roofline']['frac'], 'traffic', o['roofline'].get('traffic_measured_in_run'), o['roofline'].get('traffic_over_algorithmic'), o['roofline'].get('traffic_in_run_note'), 'fin', o['roofline_finalize']['ms_per_launch'], o['roofline_finalize'].get('traffic_measured_in_run'), o.get('tap_ms_over_fp16_headline'))
print('issue', r['roofline_issue'].get('frac'), r['roofline_issue'].get('counters_measured_in_run'), r['roofline_issue'].get('counters_in_run_note'))
PY

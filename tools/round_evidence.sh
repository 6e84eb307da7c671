#!/bin/bash
# The round's evidence in one gpurun call: the -m gpu suite, kernel stats + PMC passes of the three single-GPU configurations
# (tools/profile_round.sh), the bench lines (default form and the driver's form) and the power / clock traces.
#   gpurun --timeout 3000 -- 'bash tools/round_evidence.sh r06'        -> gpurun_out/evidence_<tag>/ + gpurun_out/profiles_<tag>/
set -u
TAG=${1:-r06}
cd "$(dirname "$0")/.."
out=gpurun_out/evidence_$TAG
mkdir -p "$out"
T0=$SECONDS
say() { echo "[evidence $((SECONDS - T0))s] $*"; }
timeout 300 python __graft_entry__.py smoke > "$out/smoke.txt" 2>&1; say "smoke: $(tail -1 "$out/smoke.txt")"
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  timeout 900 python -m pytest tests -q -m gpu > "$out/gpu_tests.txt" 2>&1
  say "pytest -m gpu: $(tail -1 "$out/gpu_tests.txt")"
  grep -E "^FAILED|^ERROR" "$out/gpu_tests.txt" | head -20
fi
[ "${SKIP_PROFILE:-0}" = 1 ] || for wl in sdxl1024 sd15 sdxl2048 sdxl1024_bf16 sdxl1024_f32acc; do
  ds=50; [ $wl = sdxl2048 ] && ds=100
  dfr=$ds; [ $wl = sdxl2048 ] && dfr=64
  nstat=20; [ $wl = sd15 ] && nstat=100      # SD-v1.5: 100 generations = the length of the headline's 20 (a 9 ms region is over before the board leaves its idle clocks)
  timeout 500 bash tools/profile_round.sh $TAG $wl $ds $dfr $nstat 4 > "$out/profile_$wl.log" 2>&1
  say "profile $wl: $(grep -c wrote "$out/profile_$wl.log") files"
done
# the bench lines read the counters measured minutes ago on this box (bench.py takes profiles/<tag>_counters.json when the kernels match)
[ -f gpurun_out/profiles_$TAG/${TAG}_counters.json ] && cp gpurun_out/profiles_$TAG/${TAG}_counters.json profiles/
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.log"; cp gpurun_out/bench_full.json "$out/bench_default_full.json"
say "bench default: $(python -c "import json;r=json.loads([l for l in open('$out/bench_default.json') if l.startswith('{')][0]);print(r['value'], r['roofline']['ms_per_launch'], r['roofline']['frac'], r['roofline'].get('traffic_over_algorithmic'), r.get('finalize_ms'), r['cpu_baseline']['value'])")"
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench_driver_form.json" 2> "$out/bench_driver_form.log"; cp gpurun_out/bench_full.json "$out/bench_driver_form_full.json"
say "bench driver form: $(python -c "import json;r=json.loads([l for l in open('$out/bench_driver_form.json') if l.startswith('{')][0]);print(r['value'], r['roofline']['ms_per_launch'], r['roofline']['frac'])")"
# power / clock while the tap launches run back to back: every byte from HBM (one step set per step) and a pool of 12 (re-used from the Infinity Cache)
[ "${SKIP_POWER:-0}" = 1 ] || for pool in 0 12; do
  O=$out/power_pool$pool; mkdir -p $O
  sample() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -E 's/^GPU\[0\][[:space:]]*:[[:space:]]*//' | tr '\n' ';'; echo; }
  echo "# python bench.py --no-baselines --no-integrated --no-other-configs --no-pmc --no-sustained --steps 3000 --warmup 10 --pool $pool   (pool 0 = one distinct step set per denoising step)" > $O/power_sclk.txt
  python bench.py --no-baselines --no-integrated --no-other-configs --no-pmc --no-sustained --steps 3000 --warmup 10 --pool $pool > $O/bench.json 2> /dev/null &
  BP=$!
  while kill -0 $BP 2>/dev/null; do echo "$(date +%s.%N) $(sample)" >> $O/power_sclk.txt; sleep 0.3; done
  python -c "
import json; d=json.loads([l for l in open('$O/bench.json') if l.startswith('{')][0]); print('# bench line of this run: maps/s', d['value'], 'tap ms', d['roofline']['ms_per_launch'], 'sustained tap ms', d.get('sustained_tap_ms'))" >> $O/power_sclk.txt
  say "power pool $pool: $(grep -c Power $O/power_sclk.txt) samples; $(tail -1 $O/power_sclk.txt | cut -c1-160)"
done
say done

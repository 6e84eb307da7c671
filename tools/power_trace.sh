#!/bin/bash
# Board power and shader clock next to the bench's tap launches (the power-limit claim needs a trace):
# rocm-smi sampled every ~0.3 s while `bench.py` runs 5000 generations back to back (~11 s), then idle.  Output: one line per sample.
R=$(pwd); O=${1:-gpurun_out/power}; mkdir -p $O
sample() { rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | sed -E 's/^GPU\[0\][[:space:]]*:[[:space:]]*//' | tr '\n' ';'; echo; }
echo "# idle" > $O/power_sclk.txt
for i in 1 2 3; do echo "$(date +%s.%N) $(sample)" >> $O/power_sclk.txt; sleep 0.3; done
echo "# under python bench.py --no-baselines --no-integrated --no-other-configs --steps 5000 --warmup 10 (tap launches back to back)" >> $O/power_sclk.txt
python bench.py --no-baselines --no-integrated --no-other-configs --steps 5000 --warmup 10 > $O/bench_power.json 2> $O/bench_power.err &
BP=$!
while kill -0 $BP 2>/dev/null; do echo "$(date +%s.%N) $(sample)" >> $O/power_sclk.txt; sleep 0.3; done
echo "# idle again" >> $O/power_sclk.txt
for i in 1 2 3; do echo "$(date +%s.%N) $(sample)" >> $O/power_sclk.txt; sleep 0.3; done
python -c "
import json; d=json.load(open('$O/bench_power.json')); print('# bench line of this run: maps/s', d['value'], 'tap ms', d['roofline']['ms_per_launch'], 'clock monitor', d['roofline_issue']['clock'])" >> $O/power_sclk.txt
cat $O/power_sclk.txt | head -60

#!/bin/bash
# round 2, GPU session 1: new parity tests, issue-rate micro-benchmark, default bench, clock / power probes
mkdir -p gpurun_out/s1
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
(ls /sys/class/drm/ ; for d in /sys/class/drm/card*/device/hwmon/hwmon*; do echo "== $d"; ls $d; for f in $d/freq1_input $d/power1_average $d/power1_input $d/power1_cap; do [ -r $f ] && echo "$f: $(cat $f)"; done; done; for f in /sys/class/drm/card*/device/pp_dpm_sclk; do echo "== $f"; cat $f; done) > gpurun_out/s1/sysfs.txt 2>&1
(rocm-smi --showclocks --showpower 2>&1 | head -40; amd-smi metric --clock --power 2>&1 | head -60) > gpurun_out/s1/smi.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/s1/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/s1/pytest.log
timeout 300 tools/ubench_issue > gpurun_out/s1/ubench_issue.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err
tail -3 gpurun_out/s1/pytest.log

#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/s17
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 600 -k "golden" > gpurun_out/s17/pytest.log 2>&1
T0=$(date +%s.%N)
timeout 900 python bench.py > gpurun_out/s17/bench_sdxl1024.json 2> gpurun_out/s17/bench_sdxl1024.err
T1=$(date +%s.%N)
echo "bench wall s: $(echo "$T1 - $T0" | bc)" > gpurun_out/s17/bench_wall.txt
bash tools/profile_round.sh r02 sdxl2048 100 24 4 2 > gpurun_out/s17/prof_sdxl2048.log 2>&1
DAAM_HIP_LIB= timeout 100 python tools/fin_sweep.py > gpurun_out/s17/fin_sweep.txt 2>&1
tail -2 gpurun_out/s17/pytest.log; cat gpurun_out/s17/bench_wall.txt
python -c "
import json
d=json.load(open('gpurun_out/s17/bench_sdxl1024.json'))
for k in ['value','ms_per_step','roofline','roofline_issue','roofline_finalize','roofline_finalize_issue','integrated']: print(k, json.dumps(d.get(k))[:700])
"

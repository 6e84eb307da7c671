#!/bin/bash
# round 6, GPU call 8: finalize time against n_rows, soak of the three finalize schedules (alone and beside two busy processes)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/exp/fin_rows_timing.py 2>&1 | tail -5
timeout 900 python tools/exp/fin_soak.py --calls 6000 2>&1 | tail -5
timeout 1200 python tools/exp/fin_soak.py --calls 3000 --noise 2 2>&1 | tail -5

#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
bash tools/power_trace.sh gpurun_out/s29

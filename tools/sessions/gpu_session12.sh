#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/s12
bash tools/profile_round.sh r02 sdxl1024 50 50 30 5 > gpurun_out/s12/prof_sdxl1024.log 2>&1
bash tools/profile_round.sh r02 sd15 50 50 30 5 > gpurun_out/s12/prof_sd15.log 2>&1
bash tools/profile_round.sh r02 sdxl2048 100 22 4 2 > gpurun_out/s12/prof_sdxl2048.log 2>&1
ls gpurun_out/profiles_r02; tail -4 gpurun_out/s12/*.log | cut -c1-300

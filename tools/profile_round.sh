#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench, three separate PMC passes, condensed into profiles/.
# usage: tools/profile_round.sh <tag> <traffic key>      e.g.  tools/profile_round.sh r01 sdxl1024:defer50:exact
set -u
TAG=${1:-r01}; KEY=${2:-sdxl1024:defer50:exact}
R=$(pwd); O=$R/gpurun_out/prof_$TAG
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
BENCH="python $R/bench.py --no-baselines --steps 40 --warmup 2"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $BENCH > $O/stats.log 2>&1
PM="python $R/bench.py --no-baselines --steps 6 --warmup 2"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $PM > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $PM > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $PM > $O/pmc_write.log 2>&1
cd $R
python tools/summarize_profiles.py --tag $TAG --stats $O/stats --pmc $O/pmc_sq $O/pmc_fetch $O/pmc_write --key $KEY --out gpurun_out/profiles_$TAG
for f in $O/*.log; do tail -n 2 $f | cut -c1-200; done
rm -rf $O/stats $O/pmc_sq $O/pmc_fetch $O/pmc_write      # raw traces are > 64 MiB: only the summaries travel back

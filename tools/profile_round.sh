#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root): kernel-trace stats of the bench, separate PMC passes
# (SQ issue counters; MFMA counters; FETCH_SIZE; WRITE_SIZE -- never combined with trace domains other than --kernel-trace),
# condensed by tools/summarize_profiles.py into gpurun_out/profiles_<tag>/ (copy what is to be judged into profiles/).
# usage: tools/profile_round.sh <tag> [workload [denoise steps [defer]]]     e.g.  tools/profile_round.sh r02 sdxl1024 50 50
set -u
TAG=${1:-r03}; WL=${2:-sdxl1024}; DS=${3:-50}; DEFER=${4:-$DS}; NSTAT=${5:-30}; NPMC=${6:-5}
ACC=exact; [ "$WL" = sdxl1024_f32acc ] && ACC=float32
KEY=$WL:defer$DEFER:$ACC
R=$(pwd); O=$R/gpurun_out/prof_${TAG}_$WL
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
ARGS="--no-baselines --no-integrated --no-other-configs --no-pmc --no-sustained --workload $WL --denoise-steps $DS"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py $ARGS --steps $NSTAT --warmup 2 > $O/stats.log 2>&1
PM="python $R/bench.py $ARGS --steps $NPMC --warmup 2"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -- $PM > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU --output-format csv -d $O/pmc_mfma -- $PM > $O/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $PM > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $PM > $O/pmc_write.log 2>&1
cd $R
# gpurun_out/ does not travel TO the box: start from the committed per-round files so that the per-workload entries of one
# build accumulate (summarize_profiles.py starts a new file when the kernel sources changed)
mkdir -p gpurun_out/profiles_$TAG
for f in ${TAG}_counters.json hbm_traffic.json; do
    [ -f gpurun_out/profiles_$TAG/$f ] || { [ -f profiles/$f ] && cp profiles/$f gpurun_out/profiles_$TAG/$f; }
done
python tools/summarize_profiles.py --tag ${TAG}_$WL --stats $O/stats --pmc $O/pmc_sq $O/pmc_mfma $O/pmc_fetch $O/pmc_write --key $KEY --out gpurun_out/profiles_$TAG
for f in $O/*.log; do tail -n 2 $f | cut -c1-200; done
rm -rf $O/stats $O/pmc_sq $O/pmc_mfma $O/pmc_fetch $O/pmc_write      # raw traces are > 64 MiB: only the summaries travel back

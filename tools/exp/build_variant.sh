#!/bin/bash
# investigation build of libdaam_hip.so (never the product):
#   build_variant.sh <out.so> [--lab] [<csrc file>=<replacement.hip> ...] [-- extra hipcc flags]
# --lab: the kernel sources with the laboratory switches of rounds 1-5 put back (timing stamps, -DDAAM_TAP_ABLATE=n, -DDAAM_TAP_TOUCH=n,
#        the Q2 / head-minor / register-staged / four-wave forms, -DDAAM_ATTEND_QLDS=1, -DDAAM_SLAB_TIMING, -DDAAM_PIPE_TIMING ...): round 6 took
#        them out of daam_amd/csrc; tools/exp/patches/lab_switches_r6.patch is that removal, applied here IN REVERSE to a scratch copy of
#        the sources as of the commit that made it (`git show <commit>:daam_amd/csrc/...`; later edits of the product kernels may make
#        the patch fail -- then check that commit out).
# (the old form `build_variant.sh <pipe src> <out.so> [flags]` = replace daam_finalize_pipe.hip)
set -e
cd "$(dirname "$0")/../.."
C=daam_amd/csrc
declare -A REP
LAB=0
if [[ "$1" == *.hip ]]; then REP[daam_finalize_pipe.hip]=$1; OUT=$2; shift 2; FLAGS=("$@")
else
  OUT=$1; shift; FLAGS=()
  while [[ $# -gt 0 ]]; do
    if [[ "$1" == "--" ]]; then shift; FLAGS=("$@"); break; fi
    if [[ "$1" == "--lab" ]]; then LAB=1; shift; continue; fi
    REP[${1%%=*}]=${1#*=}; shift
  done
fi
if [[ $LAB == 1 ]]; then
  T=$(mktemp -d /tmp/daam_lab.XXXXXX)
  mkdir -p $T/daam_amd $T/include
  cp -r $C $T/daam_amd/csrc; cp include/daam_hip.h $T/include/
  (cd $T && patch -R -p1 --no-backup-if-mismatch < "$OLDPWD/tools/exp/patches/lab_switches_r6.patch")
  C=$T/daam_amd/csrc
fi
SRCS=()
for f in daam_api.hip daam_kernels.hip daam_tap_mfma.hip daam_tap_d64.hip daam_tap_wide.hip daam_tap_chunk.hip daam_tap_slab.hip daam_attend_d64.hip daam_finalize.hip daam_finalize_pipe.hip; do
  if [[ -n "${REP[$f]}" ]]; then SRCS+=("${REP[$f]}"); else SRCS+=("$C/$f"); fi
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -I$C "${FLAGS[@]}" "${SRCS[@]}" -o $OUT
echo built $OUT

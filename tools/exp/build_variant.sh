#!/bin/bash
# investigation build: the library with some kernel files replaced by copies under tools/exp:
#   build_variant.sh <out.so> [<csrc file>=<replacement.hip> ...] [-- extra hipcc flags]
# (the old form `build_variant.sh <pipe src> <out.so> [flags]` = replace daam_finalize_pipe.hip)
set -e
cd "$(dirname "$0")/../.."
C=daam_amd/csrc
declare -A REP
if [[ "$1" == *.hip ]]; then REP[daam_finalize_pipe.hip]=$1; OUT=$2; shift 2; FLAGS=("$@")
else
  OUT=$1; shift; FLAGS=()
  while [[ $# -gt 0 ]]; do
    if [[ "$1" == "--" ]]; then shift; FLAGS=("$@"); break; fi
    REP[${1%%=*}]=${1#*=}; shift
  done
fi
SRCS=()
for f in daam_api.hip daam_kernels.hip daam_tap_mfma.hip daam_tap_d64.hip daam_tap_wide.hip daam_attend_d64.hip daam_finalize.hip daam_finalize_pipe.hip; do
  if [[ -n "${REP[$f]}" ]]; then SRCS+=("${REP[$f]}"); else SRCS+=("$C/$f"); fi
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -I$C "${FLAGS[@]}" "${SRCS[@]}" -o $OUT
echo built $OUT

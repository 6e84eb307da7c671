#!/bin/bash
# investigation build: the library with tools/exp/<src> in place of daam_finalize_pipe.hip:  build_variant.sh <src.hip> <out.so> [flags...]
set -e
cd "$(dirname "$0")/../.."
C=daam_amd/csrc
SRC=$1; OUT=$2; shift 2
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -I$C "$@" \
  $C/daam_api.hip $C/daam_kernels.hip $C/daam_tap_mfma.hip $C/daam_tap_d64.hip $C/daam_tap_wide.hip $C/daam_attend_d64.hip $C/daam_finalize.hip \
  $SRC -o $OUT
echo built $OUT

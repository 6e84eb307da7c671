"""Build ``libdaam_hip.so`` in-tree with hipcc for gfx950:  ``python -m daam_amd.build``."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['daam_api.hip', 'daam_kernels.hip', 'daam_tap_mfma.hip', 'daam_tap_d64.hip', 'daam_finalize.hip']
HEADERS = ['daam_types.h', 'daam_tap_common.h', os.path.join('..', '..', 'include', 'daam_hip.h')]
OUT = os.path.join(HERE, 'libdaam_hip.so')


def hipcc() -> str:
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found (need ROCm with gfx950 support)')
    return exe


def stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(HERE, 'csrc', f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not stale():
        return OUT
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           *[os.path.join(HERE, 'csrc', f) for f in SOURCES], '-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUT)

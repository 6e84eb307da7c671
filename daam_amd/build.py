"""Build ``libdaam_hip.so`` in-tree with hipcc for gfx950:  ``python -m daam_amd.build``."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['daam_api.hip', 'daam_kernels.hip', 'daam_tap_mfma.hip', 'daam_tap_d64.hip', 'daam_tap_wide.hip', 'daam_attend_d64.hip', 'daam_finalize.hip', 'daam_finalize_pipe.hip']
HEADERS = ['daam_types.h', 'daam_tap_common.h', 'daam_tap16.h', 'daam_tap16_softmax.h', 'daam_finalize_pipe_asm_r16.inc', 'daam_finalize_pipe_prefill_r16.inc', 'daam_finalize_pipe_asm_r8.inc', 'daam_finalize_pipe_prefill_r8.inc', os.path.join('..', '..', 'include', 'daam_hip.h')]
OUT = os.path.join(HERE, 'libdaam_hip.so')


def csrc_sha() -> str:
    """Fingerprint of the kernel sources (csrc/*.hip, *.h and the C header): ties committed profiler summaries
    (profiles/*.json) to the build they were measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + HEADERS):
        with open(os.path.join(HERE, 'csrc', f), 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


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


def fastpath_out() -> str:
    import sysconfig
    return os.path.join(HERE, '_fastpath' + sysconfig.get_config_var('EXT_SUFFIX'))


def build_fastpath(force: bool = False, verbose: bool = True) -> str:
    """``daam_amd._fastpath``: the host-side tap recorder (csrc/daam_fastpath.cpp), a CPython extension
    compiled against the torch headers of the running interpreter (host code only, no device code)."""
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ext
    src = os.path.join(HERE, 'csrc', 'daam_fastpath.cpp')
    out = fastpath_out()
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    lib_dirs = ext.library_paths()
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    # host code only; the HIP platform define and headers are for c10::hip::getCurrentHIPStream (the recorder launches
    # daam_attend on torch's current stream), nothing here is compiled for the device
    cmd = ['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-D__HIP_PLATFORM_AMD__', '-DUSE_ROCM',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}',
           '-I' + sysconfig.get_paths()['include'], *['-I' + p for p in ext.include_paths()], '-I' + os.path.join(rocm, 'include'),
           src, '-o', out, *['-L' + p for p in lib_dirs], *['-Wl,-rpath,' + p for p in lib_dirs],
           '-ltorch_python', '-ltorch', '-ltorch_cpu', '-lc10', '-lc10_hip']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


def build_variant(out: str, flags, verbose: bool = True) -> str:
    """A/B builds for kernel experiments (tools/): the same sources with extra ``-D`` flags into another file; load it with
    ``DAAM_HIP_LIB=<out>``."""
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fvisibility=hidden', *flags,
           *[os.path.join(HERE, 'csrc', f) for f in SOURCES], '-o', out]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    build_fastpath(force, verbose)
    if not force and not stale():
        return OUT
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fvisibility=hidden',
           *[os.path.join(HERE, 'csrc', f) for f in SOURCES], '-o', OUT]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUT)

"""Build ``libdaam_hip.so`` in-tree with hipcc for gfx950:  ``python -m daam_amd.build``."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['daam_api.hip', 'daam_kernels.hip', 'daam_tap_mfma.hip', 'daam_tap_d64.hip', 'daam_tap_wide.hip', 'daam_tap_chunk.hip', 'daam_tap_slab.hip', 'daam_attend_d64.hip', 'daam_finalize.hip', 'daam_finalize_pipe.hip']
HEADERS = ['daam_types.h', 'daam_tap_common.h', 'daam_tap16.h', 'daam_tap16_softmax.h', 'daam_finalize_pipe_asm_r16.inc', 'daam_finalize_pipe_prefill_r16.inc', os.path.join('..', '..', 'include', 'daam_hip.h')]
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


def kernel_shas(lib: str = OUT) -> dict:
    """``{mangled kernel name: fingerprint of its machine code}`` for every kernel in a built library: the gfx950 code objects
    are cut out of the offload bundles in ``.hip_fatbin`` and the bytes of every function symbol hashed.  Unlike ``csrc_sha``
    this does not change when a source file gains an ``#if``-guarded experiment or when another kernel file is added, and it
    does change when the compiler emits different code for an untouched source: it is what ties a committed profiler summary to
    the kernels of THIS build (bench.py ``load_counters``)."""
    import hashlib
    import struct
    data = open(lib, 'rb').read()
    magic = b'__CLANG_OFFLOAD_BUNDLE__'
    out = {}
    pos = data.find(magic)
    while pos >= 0:
        n, = struct.unpack_from('<Q', data, pos + len(magic))
        o = pos + len(magic) + 8
        for _ in range(n):
            off, size, ts = struct.unpack_from('<QQQ', data, o)
            o += 24
            triple = data[o:o + ts].decode()
            o += ts
            if 'gfx950' in triple and size:
                out.update(_elf_function_shas(data[pos + off:pos + off + size], hashlib))
        pos = data.find(magic, pos + 1)
    return out


def _elf_function_shas(elf: bytes, hashlib) -> dict:
    import struct
    if elf[:4] != b'\x7fELF' or elf[4] != 2:
        return {}
    shoff, = struct.unpack_from('<Q', elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from('<HHH', elf, 0x3A)
    secs = [struct.unpack_from('<IIQQQQIIQQ', elf, shoff + i * shentsize) for i in range(shnum)]
    out = {}
    for (_, typ, _, _, off, size, link, _, _, entsize) in secs:
        if typ != 2:                                         # SHT_SYMTAB
            continue
        str_off = secs[link][4]
        for i in range(size // entsize):
            name_i, info, _, shndx, value, fsize = struct.unpack_from('<IBBHQQ', elf, off + i * entsize)
            if (info & 0xF) != 2 or not fsize or shndx >= len(secs):      # STT_FUNC with a body
                continue
            end = elf.index(b'\0', str_off + name_i)
            name = elf[str_off + name_i:end].decode()
            sec = secs[shndx]
            start = sec[4] + (value - sec[3])
            out[name] = hashlib.sha256(elf[start:start + fsize]).hexdigest()[:12]
    return out


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


def _deps_of(src: str) -> list:
    """Headers / generated includes a source depends on (every HEADERS entry: they are few and cheap to stat)."""
    return [os.path.join(HERE, 'csrc', src)] + [os.path.join(HERE, 'csrc', h) for h in HEADERS]


def build(force: bool = False, verbose: bool = True, jobs: int = 0) -> str:
    """One object per source (``daam_amd/build/*.o``, compiled in parallel, only when the source or a header is newer), one link.
    The device code of a source is the same whether it is compiled alone or in one hipcc command with the others (no -fgpu-rdc):
    ``kernel_shas()`` of the library is what ties profiles to a build."""
    from concurrent.futures import ThreadPoolExecutor
    build_fastpath(force, verbose)
    if not force and not stale():
        return OUT
    objdir = os.path.join(HERE, 'build')
    os.makedirs(objdir, exist_ok=True)
    flags = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fvisibility=hidden']
    todo = []
    for f in SOURCES:
        obj = os.path.join(objdir, f + '.o')
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in _deps_of(f)):
            todo.append([hipcc(), *flags, '-c', os.path.join(HERE, 'csrc', f), '-o', obj])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 1)) as ex:
        list(ex.map(run, todo))
    run([hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-fvisibility=hidden',
         *[os.path.join(objdir, f + '.o') for f in SOURCES], '-o', OUT])
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(OUT)

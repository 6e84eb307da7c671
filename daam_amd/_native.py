"""ctypes binding of ``libdaam_hip.so`` (C ABI: ``include/daam_hip.h``).

There is no CPU or PyTorch fallback: if the HIP library is missing or no MI355X is visible
the import / first call fails loudly.  Build the library with ``python -m daam_amd.build``
(or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8, c_void_p

LIB_NAME = 'libdaam_hip.so'
LIB_PATH = os.environ.get('DAAM_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

ABI_VERSION = 6
DAAM_F16, DAAM_F32, DAAM_BF16 = 0, 1, 2
E_INVALID, E_STATE, E_NOMAPS, E_UNSUPPORTED = -1, -2, -3, -4

# every symbol include/daam_hip.h declares (tests check the library exports exactly these)
EXPORTS = (
    'daam_abi_version', 'daam_last_error', 'daam_ctx_create', 'daam_ctx_destroy', 'daam_layer_configure',
    'daam_layer_acc', 'daam_layer_touch', 'daam_layer_release', 'daam_reset', 'daam_tap_qk', 'daam_tap_qk_enqueue', 'daam_tap_qk_enqueue_many', 'daam_tap_pending', 'daam_tap_flush',
    'daam_tap_probs', 'daam_attend_supported', 'daam_attend', 'daam_key_offset', 'daam_finalize', 'daam_finalize_prepare', 'daam_epilogue_normalize', 'daam_word_heat_map', 'daam_mask_overlap',
    'daam_last_launch', 'daam_last_flush', 'daam_last_kernels', 'daam_profile_enable', 'daam_profile_last_ms', 'daam_profile_history', 'daam_clock_monitor_start', 'daam_clock_monitor_read',
)


class DaamError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f'libdaam_hip error {code}: {message}')
        self.code = code


class QKDesc(Structure):
    """``DaamQKDesc`` (include/daam_hip.h)."""
    _fields_ = [
        ('in_dtype', c_int32), ('batch', c_int32), ('heads', c_int32), ('hw', c_int32), ('tokens', c_int32),
        ('head_dim', c_int32), ('round_logits', c_int32), ('scale', c_float),
        ('q_stride_b', c_int64), ('q_stride_h', c_int64), ('q_stride_p', c_int64),
        ('k_stride_b', c_int64), ('k_stride_h', c_int64), ('k_stride_t', c_int64),
    ]


class AttendDesc(Structure):
    """``DaamAttendDesc`` (include/daam_hip.h)."""
    _fields_ = [
        ('qk', QKDesc),
        ('v_stride_b', c_int64), ('v_stride_h', c_int64), ('v_stride_t', c_int64),
        ('o_stride_b', c_int64), ('o_stride_h', c_int64), ('o_stride_p', c_int64),
    ]


_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} is missing: the MI355X heat-map path has no fallback. '
            f'Build it with `python -m daam_amd.build` (needs hipcc, --offload-arch=gfx950).')
    lib = ctypes.CDLL(LIB_PATH)
    lib.daam_abi_version.restype = c_int
    lib.daam_last_error.restype = c_char_p
    lib.daam_ctx_create.argtypes = [c_int, c_int, c_int, c_int, POINTER(c_void_p)]
    lib.daam_ctx_destroy.argtypes = [c_void_p]
    lib.daam_layer_configure.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.daam_layer_acc.argtypes = [c_void_p, c_int, POINTER(c_void_p), POINTER(c_size_t)]
    lib.daam_layer_touch.argtypes = [c_void_p, c_int, c_void_p]
    lib.daam_layer_release.argtypes = [c_void_p, c_int]
    lib.daam_reset.argtypes = [c_void_p, c_void_p]
    lib.daam_tap_qk.argtypes = [c_void_p, c_int, c_void_p, c_void_p, POINTER(QKDesc), c_void_p]
    lib.daam_tap_qk_enqueue.argtypes = [c_void_p, c_int, c_void_p, c_void_p, POINTER(QKDesc)]
    lib.daam_tap_qk_enqueue_many.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]
    lib.daam_tap_pending.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int)]
    lib.daam_tap_flush.argtypes = [c_void_p, c_void_p]
    lib.daam_tap_probs.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.daam_attend_supported.argtypes = [POINTER(AttendDesc), c_void_p, c_void_p, c_void_p, c_void_p]
    lib.daam_attend.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(AttendDesc), c_int, c_void_p]
    lib.daam_key_offset.argtypes = [c_void_p, c_int, POINTER(c_int), POINTER(c_int)]
    lib.daam_finalize.argtypes = [c_void_p, POINTER(c_uint8), c_int, c_void_p, c_void_p]
    lib.daam_finalize_prepare.argtypes = [c_void_p, POINTER(c_uint8), c_int, c_void_p, c_void_p]
    lib.daam_epilogue_normalize.argtypes = [c_void_p, c_int, c_int, c_void_p]
    lib.daam_word_heat_map.argtypes = [c_void_p, c_int, POINTER(c_int32), c_int, c_void_p, c_void_p, c_int, c_int,
                                       c_int, c_float, c_void_p, c_void_p]
    lib.daam_mask_overlap.argtypes = [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.daam_profile_history.argtypes = [c_void_p, c_int, POINTER(c_float), c_int, POINTER(c_int)]
    lib.daam_last_launch.argtypes = [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    lib.daam_last_flush.argtypes = [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(ctypes.c_longlong)]
    lib.daam_last_kernels.argtypes = [c_void_p, c_int, ctypes.c_char_p, c_int]
    lib.daam_profile_enable.argtypes = [c_void_p, c_int]
    lib.daam_profile_last_ms.argtypes = [c_void_p, c_int, POINTER(c_float)]
    lib.daam_clock_monitor_start.argtypes = [c_void_p, c_int, c_int]
    lib.daam_clock_monitor_read.argtypes = [c_void_p, POINTER(c_float), c_int, POINTER(c_int)]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name not in ('daam_last_error',):
            fn.restype = c_int
    if lib.daam_abi_version() != ABI_VERSION:
        raise RuntimeError(f'{LIB_PATH}: ABI version {lib.daam_abi_version()} != {ABI_VERSION}; rebuild')
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().daam_last_error()
        raise DaamError(rc, msg.decode() if msg else '')


__all__ = ['load', 'check', 'DaamError', 'QKDesc', 'AttendDesc', 'DAAM_F16', 'DAAM_F32', 'DAAM_BF16', 'EXPORTS', 'LIB_PATH',
           'byref', 'c_void_p', 'c_int', 'c_uint8', 'c_int32', 'c_size_t', 'E_NOMAPS']

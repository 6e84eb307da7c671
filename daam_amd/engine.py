"""Device-side state of one trace: the per-(layer, head) running sums and the native context.

Python owns the memory (torch tensors, so ``all_heat_maps`` hands out zero-copy views and the
caching allocator sees it); ``libdaam_hip.so`` does all arithmetic.  Mirrors what the
reference keeps in ``RawHeatMapCollection`` (daam/heatmap.py:148-172) plus the bodies of
``UNetCrossAttentionHooker.__call__`` between scores and bmm (daam/trace.py:285-294) and
``compute_global_heat_map`` (daam/trace.py:103-130).
"""
from __future__ import annotations

import ctypes
import math
import os
import sys
import weakref
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as nat

Key = Tuple[int, int, int]   # (factor, layer, head) -- daam/heatmap.py:145


class CallShape:
    """What a layer's last validated ``tap_qk`` / ``attend`` call looked like, with the descriptor built for it -- the
    per-layer cache entry of the engine.  ``csrc/daam_fastpath.cpp`` mirrors it field by field in ``struct CallShape``
    (``Recorder.set_cache`` / ``set_attend_cache`` copy an entry across)."""
    __slots__ = ('q_shape', 'k_shape', 'dtype', 'heads', 'scale', 'round_logits', 'factor', 'desc_ref', 'desc', 'desc_addr',
                 'q_numel', 'k_numel', 'held_bytes')

    def __init__(self, query, key, heads, scale, round_logits, factor, desc):
        self.q_shape, self.k_shape, self.dtype = query.shape, key.shape, query.dtype
        self.heads, self.scale, self.round_logits, self.factor = heads, scale, round_logits, factor
        self.desc = desc                                       # DaamQKDesc / DaamAttendDesc (kept alive here), or None
        self.desc_ref = nat.byref(desc) if desc is not None else None
        self.desc_addr = ctypes.addressof(desc) if desc is not None else 0
        self.q_numel, self.k_numel = query.numel(), key.numel()
        self.held_bytes = (self.q_numel + self.k_numel) * query.element_size()   # what a recorded call keeps alive

    def same_call(self, query, key, heads, scale, round_logits, factor=None) -> bool:
        return (self.q_shape == query.shape and self.k_shape == key.shape and self.dtype is query.dtype
                and key.dtype is self.dtype and self.heads == heads and self.scale == scale
                and self.round_logits == round_logits and (factor is None or self.factor == factor))

# pipeline / running-sum dtypes the library knows (include/daam_hip.h).  fp16 runs on the MFMA kernels;
# bf16 and fp32 on the any-shape kernels with the same rounding points.
_DTYPE_CODE = {torch.float16: nat.DAAM_F16, torch.float32: nat.DAAM_F32, torch.bfloat16: nat.DAAM_BF16}


# Parked native contexts (with their running-sum buffers) of closed engines, per (device, layers, tokens, map side,
# sum dtype).  A pipeline is usually traced once per generation: re-adopting the previous trace's context saves the
# 1.5 ms of set-up / tear-down and the device synchronisation that destroying a context implies (hipFree).
_PARKED: Dict[tuple, list] = {}
_PARK_LIMIT = 2


_FASTPATH_READY = False


def _load_fastpath():
    """The C++ recorder, or None when the extension is not built.  Its release thread (the storages of the recorded
    Q / K go back to the caching allocator off the interpreter's thread) is drained and stopped at interpreter exit,
    while torch is still intact; ``DAAM_SYNC_RELEASE=1`` frees inline instead."""
    global _FASTPATH_READY
    try:
        from . import _fastpath
    except ImportError:
        return None
    if not _FASTPATH_READY:
        import atexit
        atexit.register(_fastpath.shutdown)
        if os.environ.get('DAAM_SYNC_RELEASE'):
            _fastpath.set_sync_release(True)
        _FASTPATH_READY = True
    return _fastpath


def drain_released() -> None:
    """Block until every Q / K block released by a finished launch is back in the caching allocator (they are freed on
    a helper thread).  Only memory accounting needs this (``torch.cuda.memory_allocated`` right after a trace)."""
    fp = _load_fastpath()
    if fp is not None:
        fp.drain()


def release_parked_contexts() -> None:
    """Destroy every parked context now (frees their sum buffers too).  Not called at interpreter exit on purpose:
    process teardown reclaims them, and no HIP call has to run while the runtime is shutting down."""
    for states in _PARKED.values():
        for st in states:
            st['lib'].daam_ctx_destroy(st['ctx'])        # synchronises the device first
    _PARKED.clear()


class HeatMapEngine:
    def __init__(self, n_layers: int, tokens: int = 77, out_side: int = 64, accumulate: str = 'exact',
                 defer_steps: int = 0, defer_bytes: int = 32 << 30, reuse_context: bool = False):
        """``accumulate``: ``'exact'`` keeps the running sums in the pipeline dtype like the
        reference (fp16 sums on an fp16 pipeline, heatmap.py:156); ``'float32'`` is the
        accuracy mode.  ``defer_steps`` > 0 records Q/K pointers and taps ``defer_steps``
        denoising steps of all layers in one launch (at most 64); the Q / K of the recorded steps stay
        alive until then, and a launch is forced at the next step boundary once they add up to
        ``defer_bytes`` (both CFG halves count: 388 MB per SDXL-1024 step).  ``reuse_context``: ``close()`` parks
        the native context and the sum buffers for the next engine of the same geometry instead of destroying them
        (what ``trace`` asks for: one trace per generation is the normal use)."""
        if accumulate not in ('exact', 'float32'):
            raise ValueError("accumulate must be 'exact' or 'float32'")
        self.lib = nat.load()
        self.n_layers = int(n_layers)
        self.tokens = int(tokens)
        self.out_side = int(out_side)
        self.accumulate = accumulate
        self.defer_steps = min(int(defer_steps), 64)     # the kernels stage at most 64 steps of pointers per launch
        self.defer_bytes = int(defer_bytes) if defer_bytes and defer_bytes > 0 else 1 << 62
        self.reuse_context = bool(reuse_context) and not os.environ.get('DAAM_NO_CTX_POOL')
        # debugging aid: remember tensor._version of every recorded Q / K and refuse to launch when one was written in
        # place between the processor call and the launch (deferred taps read them at launch time)
        self._check_versions = bool(os.environ.get('DAAM_CHECK_VERSIONS'))
        self._rec_stream: Optional[torch.cuda.Stream] = None   # stream the generation's Q / K are produced on
        self._views_out = False                           # all_heat_maps handed out views of the live sum buffers
        self._held = 0                                    # bytes of recorded Q / K (Python recorder)
        self.ctx: Optional[nat.c_void_p] = None
        self.device: Optional[torch.device] = None
        self.acc_dtype: Optional[torch.dtype] = None
        self.acc: Dict[int, torch.Tensor] = {}           # layer -> [heads, tokens, side, side]
        self.layer_info: Dict[int, Tuple[int, int, int]] = {}   # layer -> (factor, heads, side)
        self.touched: List[int] = []                     # layers updated since clear(), first-update order
        # deferred taps: recorded per call (the tensors are kept alive until the flush)
        self._rec: List[tuple] = []                      # (layer, query, key, address of its DaamQKDesc)
        self._window = self.defer_steps                  # steps of one layer per launch
        self._cnt: List[int] = [0] * self.n_layers      # recorded steps per layer
        self._qk_cache: List[Optional[CallShape]] = [None] * self.n_layers
        self._att_cache: List[Optional[CallShape]] = [None] * self.n_layers   # attend(): per-layer call descriptors
        self._touched_flag: List[bool] = [False] * self.n_layers
        self._mask_cache: Dict[tuple, tuple] = {}        # finalize key masks per selection
        # deferred mode: the per-call bookkeeping runs in the C++ recorder (csrc/daam_fastpath.cpp) when
        # that extension is built; the Python implementation below is the same logic and stays the
        # slow path (first call of a layer, shape changes) and the fallback.  Both only record host-side
        # pointers -- all arithmetic is in libdaam_hip either way.
        self._fast = None
        if self.defer_steps and not os.environ.get('DAAM_NO_FASTPATH') and not self._check_versions:
            _fastpath = _load_fastpath()
            if _fastpath is not None:
                # the recorder calls back into this engine through a weak reference: engine -> recorder is the
                # only strong edge, so dropping the trace frees the context and the running sums at once
                # (no reference cycle waiting for the garbage collector with 221 MB of sums attached)
                me = weakref.ref(self)
                self._fast = _fastpath.Recorder(self.n_layers,
                                                lambda *a, **k: me()._tap_qk_slow(*a, **k),
                                                lambda: me().flush(),
                                                lambda layer: me()._touch(layer))
                self._fast.set_window(self._window)
                self._fast.set_budget(self.defer_bytes)
                self.tap_qk = self._fast.tap
                # daam_attend from C++ too (the recorder gets the context and the entry point once the context exists)
                self._attend_addr = 0
                try:
                    self._attend_addr = ctypes.cast(self.lib.daam_attend, ctypes.c_void_p).value or 0
                except (ctypes.ArgumentError, TypeError, AttributeError):
                    pass                                   # not a ctypes library (tests drive the engine with a recording fake)
                self.attend = self._fast.attend
                self._sync_native()

    def _sync_native(self) -> None:
        """Tell the C++ recorder which native context (if any) its ``attend`` launches on."""
        if self._fast is not None:
            me = weakref.ref(self)
            ctx = self.ctx.value if self.ctx is not None and getattr(self.ctx, 'value', None) else 0
            self._fast.set_native(ctx or 0, self._attend_addr if ctx else 0, lambda *a, **k: HeatMapEngine.attend(me(), *a, **k))    # the Python method: the recorder's slow path

    # ---- lifetime --------------------------------------------------------------------------
    def _require_device(self, t: torch.Tensor) -> None:
        if t.device.type != 'cuda':
            raise RuntimeError(
                'daam_amd: heat-map extraction runs only on an MI355X (HIP device); got a tensor on '
                f'{t.device}. There is no CPU fallback.')
        if self.device is None:
            self.device = t.device
        elif self.device != t.device:
            raise RuntimeError(f'daam_amd: trace is bound to {self.device}, got a tensor on {t.device}')

    def _ensure_ctx(self, pipe_dtype: torch.dtype) -> None:
        if self.ctx is not None:
            return
        if pipe_dtype not in _DTYPE_CODE:
            raise RuntimeError(f'daam_amd: unsupported pipeline dtype {pipe_dtype} (fp16 / bf16 / fp32 only)')
        self.acc_dtype = torch.float32 if self.accumulate == 'float32' else pipe_dtype
        parked = _PARKED.get(self._park_key()) if self.reuse_context else None
        if parked:
            st = parked.pop()
            self.ctx, self.acc, self.layer_info = st['ctx'], st['acc'], st['layer_info']
            # whatever the previous owner queued (on its stream) comes first
            self._current_stream().wait_event(st['event'])
            nat.check(self.lib.daam_reset(self.ctx, self.stream))      # sums start from zero (lazily, like clear())
            self._sync_native()
            return
        ctx = nat.c_void_p()
        with torch.cuda.device(self.device):
            nat.check(self.lib.daam_ctx_create(self.n_layers, self.tokens, self.out_side,
                                               _DTYPE_CODE[self.acc_dtype],
                                               nat.byref(ctx)))
        self.ctx = ctx
        self._sync_native()

    def _park_key(self) -> tuple:
        return (str(self.device), self.n_layers, self.tokens, self.out_side, self.acc_dtype)

    def close(self) -> None:
        if self.ctx is not None:
            # a context whose sum buffers were handed out as views (all_heat_maps iteration) is not parked: the next
            # owner would overwrite what those views show.  Destroying the context leaves the (torch-owned) buffers
            # to the views.
            park = self.reuse_context and not self._views_out
            states = _PARKED.setdefault(self._park_key(), []) if park else None
            if states is not None and len(states) < _PARK_LIMIT:
                done = self._current_stream().record_event()
                states.append(dict(lib=self.lib, ctx=self.ctx, acc=self.acc, layer_info=self.layer_info, event=done))
                self.acc, self.layer_info = {}, {}
            else:
                self.lib.daam_ctx_destroy(self.ctx)
            self.ctx = None
            self._sync_native()
        self.acc.clear()
        self.layer_info.clear()
        self.touched.clear()
        self._touched_flag = [False] * self.n_layers
        self._qk_cache = [None] * self.n_layers
        self._att_cache = [None] * self.n_layers
        if self._fast is not None:
            self._fast.invalidate()
        self._drop_recorded()

    def __del__(self):
        if sys is None or sys.is_finalizing():          # no HIP calls while the interpreter (and the HIP runtime) shut down
            return
        try:
            self.close()
        except Exception:
            pass

    def _current_stream(self) -> 'torch.cuda.Stream':
        return torch.cuda.current_stream(self.device)

    @property
    def stream(self) -> int:
        return self._current_stream().cuda_stream

    def _ensure_layer(self, layer: int, heads: int, side: int, factor: int) -> None:
        info = self.layer_info.get(layer)
        if info == (factor, heads, side):
            return
        if info is not None:
            # the reference would simply start a new key set / fail on a shape mismatch in `+`
            self.flush()
        buf = torch.zeros(heads, self.tokens, side, side, dtype=self.acc_dtype, device=self.device)
        nat.check(self.lib.daam_layer_configure(self.ctx, layer, heads, side, factor, buf.data_ptr()))
        self.acc[layer] = buf
        self.layer_info[layer] = (factor, heads, side)
        self._mask_cache.clear()

    def _touch(self, layer: int) -> None:
        if not self._touched_flag[layer]:
            if not self.touched:
                # first tap of a generation: this is the stream its Q / K are produced on (see flush)
                self._rec_stream = self._current_stream()
            self._touched_flag[layer] = True
            self.touched.append(layer)

    # ---- RawHeatMapCollection.clear (heatmap.py:170-172) -----------------------------------------
    def clear(self) -> None:
        self._drop_recorded()
        self.touched.clear()
        self._touched_flag = [False] * self.n_layers
        self._set_window(self.defer_steps)
        if self._fast is not None:
            self._fast.reset_touched()
        if self.ctx is not None:
            nat.check(self.lib.daam_reset(self.ctx, self.stream))
        if self._views_out:
            # the reference's clear() drops its dict and the tensors it handed out live on unchanged
            # (heatmap.py:170-172): leave the old buffers to those views and start the next generation on new ones.
            # The native context must forget them too -- daam_reset only marked them "to be zeroed", and a layer
            # that is not tapped again (another resolution turns its factor into 8) would otherwise be zeroed by the
            # next finalize: a write into memory the views own, or that is back in the caching allocator.
            if self.ctx is not None:
                for layer in list(self.layer_info):
                    nat.check(self.lib.daam_layer_release(self.ctx, layer))
            self.acc, self.layer_info = {}, {}
            self._qk_cache = [None] * self.n_layers
            self._att_cache = [None] * self.n_layers
            self._mask_cache.clear()
            if self._fast is not None:
                self._fast.invalidate()
            self._views_out = False

    # ---- tap -------------------------------------------------------------------------------------
    def tap_qk(self, layer: int, query: torch.Tensor, key: torch.Tensor, heads: int, scale: float,
               factor: int, round_logits: bool = True) -> None:
        """``query`` [B, hw, heads*d], ``key`` [B, tokens, heads*d] straight out of ``to_q`` /
        ``to_k`` (trace.py:262,269); no ``head_to_batch_dim`` copy is made."""
        # Hot path (3000 calls per SDXL generation).  Full validation (shapes, dtypes, call parameters)
        # runs on a layer's first call of every launch window; inside a window only the invariants
        # that could silently change the memory layout are re-checked (element counts, dtype,
        # contiguity) -- every torch attribute access costs 50-100 ns here.
        c = self._qk_cache[layer]
        cnt = self._cnt
        if (c is None or cnt[layer] == 0 or not self.defer_steps):
            if (c is None or not c.same_call(query, key, heads, scale, round_logits, factor)
                    or not query.is_contiguous() or not key.is_contiguous()):
                query, key, c = self._prepare_qk(layer, query, key, heads, scale, factor, round_logits)
        elif (query.numel() != c.q_numel or key.numel() != c.k_numel or query.dtype is not c.dtype or c.scale != scale
              or not query.is_contiguous() or not key.is_contiguous()):
            query, key, c = self._prepare_qk(layer, query, key, heads, scale, factor, round_logits)
        if self.defer_steps:
            # record only: pointers cross the FFI in one daam_tap_qk_enqueue_many call per flush
            n = cnt[layer]
            if n >= self._window or (self._held >= self.defer_bytes and self._rec[0][0] == layer):
                self.flush()
                n = 0
            cnt[layer] = n + 1
            self._held += c.held_bytes
            if self._check_versions:
                self._rec.append((layer, query, key, c.desc_addr, query._version, key._version))
            else:
                self._rec.append((layer, query, key, c.desc_addr))
        else:
            rc = self.lib.daam_tap_qk(self.ctx, layer, query.data_ptr(), key.data_ptr(), c.desc_ref, self.stream)
            if rc:
                nat.check(rc)
        if not self._touched_flag[layer]:
            self._touch(layer)

    def _tap_qk_slow(self, layer: int, query: torch.Tensor, key: torch.Tensor, heads: int, scale: float,
                     factor: int, round_logits: bool = True) -> None:
        """Called by the C++ recorder for everything but the steady state: validates, rebuilds the
        layer's call descriptor, teaches the recorder the new call shape and records the tap."""
        f = self._fast
        c = self._qk_cache[layer] if 0 <= layer < self.n_layers else None
        fresh = (c is None or not c.same_call(query, key, heads, scale, round_logits, factor)
                 or not query.is_contiguous() or not key.is_contiguous())
        if fresh:
            query, key, c = self._prepare_qk(layer, query, key, heads, scale, factor, round_logits)
        if f.full(layer):
            self.flush()
        if fresh:
            f.set_cache(layer, query, key, int(heads), float(scale), int(factor), bool(round_logits), c.desc_addr)
        f.record(layer, query, key, c.desc_addr)
        self._touch(layer)

    @property
    def pending_taps(self) -> int:
        """Recorded taps that have not been launched yet."""
        return self._fast.count() if self._fast is not None else len(self._rec)

    def last_flush(self) -> dict:
        """Launch structure of the last deferred tap launch (``daam_last_flush``): kernels launched, how many of them ran on
        auxiliary streams beside the caller's, the longest per-layer step chain, and the number of tap launches this
        context has made so far."""
        if self.ctx is None:
            return dict(kernels=0, side_streams=0, max_steps=0, launches=0)
        k, sd, ms, n = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_longlong()
        nat.check(self.lib.daam_last_flush(self.ctx, ctypes.byref(k), ctypes.byref(sd), ctypes.byref(ms), ctypes.byref(n)))
        return dict(kernels=k.value, side_streams=sd.value, max_steps=ms.value, launches=n.value)

    def last_kernels(self, which: int = 0) -> str:
        """Names of the kernel(s) the last tap launch (``which`` 0) / finalize call (1) really launched (``daam_last_kernels``)."""
        if self.ctx is None:
            return ''
        buf = ctypes.create_string_buffer(256)
        nat.check(self.lib.daam_last_kernels(self.ctx, which, buf, len(buf)))
        return buf.value.decode()

    def _pending(self, layer: int) -> int:
        return self._fast.pending(layer) if self._fast is not None else self._cnt[layer]

    def _set_window(self, w: int) -> None:
        self._window = w
        if self._fast is not None:
            self._fast.set_window(w)

    def _prepare_qk(self, layer, query, key, heads, scale, factor, round_logits):
        """Slow path of ``tap_qk``: validate, (re)configure the layer, build the call descriptor.
        Returns ``(query, key, cache entry)`` with both tensors contiguous."""
        if not 0 <= layer < self.n_layers:
            raise IndexError(f'layer {layer} out of range (trace has {self.n_layers} layers)')
        self._require_device(query)
        query = query if query.is_contiguous() else query.contiguous()
        key = key if key.is_contiguous() else key.contiguous()
        self._ensure_ctx(query.dtype)
        if query.dtype != key.dtype:
            raise RuntimeError('daam_amd: query / key dtype mismatch')
        if query.dtype not in _DTYPE_CODE:
            raise RuntimeError(f'daam_amd: unsupported activation dtype {query.dtype} (fp16 / bf16 / fp32 only)')
        if self.acc_dtype not in (torch.float32, query.dtype):
            raise RuntimeError(f'daam_amd: {query.dtype} activations on a trace whose running sums are {self.acc_dtype} '
                               '(fp32 activations need fp32 sums; fp16 / bf16 sums need activations of the same dtype)')
        b, hw, c = query.shape
        tokens = key.shape[1]
        d = c // heads
        side = int(math.sqrt(hw))
        bh = b * heads
        self._ensure_layer(layer, bh - bh // 2, side, factor)
        desc = nat.QKDesc(
            in_dtype=_DTYPE_CODE[query.dtype],
            batch=b, heads=heads, hw=hw, tokens=tokens, head_dim=d, round_logits=1 if round_logits else 0,
            scale=float(scale),
            q_stride_b=hw * c, q_stride_h=d, q_stride_p=c,
            k_stride_b=tokens * c, k_stride_h=d, k_stride_t=c)
        # a shape change of a layer inside a deferred batch starts a new batch (the C side checks too)
        if self._pending(layer):
            self.flush()
        entry = CallShape(query, key, heads, scale, round_logits, factor, desc)
        self._qk_cache[layer] = entry
        return query, key, entry

    # ---- attend: the processor's attention on the library's kernel, tap fused in -------------------
    def attend(self, layer: int, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, heads: int, scale: float,
               factor: int, round_logits: bool = True, tapped: bool = True) -> Optional[torch.Tensor]:
        """``softmax(scale * Q K^T) V`` of one cross-attention call with the reference's rounding points
        (``get_attention_scores`` + ``bmm``, daam/trace.py:276,296-297) on ``daam_attend``; returns ``[B, hw, heads*d]``
        ready for the output projection, or ``None`` when the call is not one the kernel takes (not fp16 / bf16, head_dim not a
        multiple of 8 up to 160, not 77 keys, not contiguous): the caller then uses the framework's attention and ``tap_qk``.

        ``tapped``: the call passes the reference's gate (trace.py:289).  On an immediate trace (``defer_steps=0``) the
        heat-map update happens inside the same kernel; on a deferred trace the kernel only attends and Q / K are
        recorded for the batched launch, exactly as ``tap_qk`` would."""
        a = self._att_cache[layer] if 0 <= layer < self.n_layers else None
        if (a is None or a.q_shape != query.shape or a.k_shape != key.shape or a.dtype is not query.dtype or a.heads != heads
                or a.scale != scale or a.round_logits != round_logits):
            a = self._prepare_attend(layer, query, key, value, heads, scale, round_logits)
        if (a.desc is None or value.shape != a.k_shape or key.dtype is not a.dtype or value.dtype is not a.dtype
                or not (query.is_contiguous() and key.is_contiguous() and value.is_contiguous())
                or ((query.requires_grad or key.requires_grad or value.requires_grad) and torch.is_grad_enabled())):
            return None                                                    # the kernel has no backward: leave autograd to torch
        fused_tap = tapped and not self.defer_steps
        if fused_tap:
            c = self._qk_cache[layer]
            if c is None or not c.same_call(query, key, heads, scale, round_logits, factor):
                self._prepare_qk(layer, query, key, heads, scale, factor, round_logits)    # validates, configures the layer
        out = torch.empty_like(query)
        rc = self.lib.daam_attend(self.ctx, layer, query.data_ptr(), key.data_ptr(), value.data_ptr(), out.data_ptr(),
                                  a.desc_ref, 1 if fused_tap else 0, self.stream)
        if rc:
            if rc == nat.E_UNSUPPORTED:                    # e.g. a view whose data pointer is not 16-byte aligned
                return None
            nat.check(rc)
        if fused_tap:
            if not self._touched_flag[layer]:
                self._touch(layer)
        elif tapped:
            self.tap_qk(layer, query, key, heads, scale, factor, round_logits)
        return out

    def _prepare_attend(self, layer, query, key, value, heads, scale, round_logits):
        if not 0 <= layer < self.n_layers:
            raise IndexError(f'layer {layer} out of range (trace has {self.n_layers} layers)')
        self._require_device(query)
        desc = None
        d = query.shape[2] // heads if query.dim() == 3 and heads > 0 else 0
        # bf16 pipelines: the kernel rounds the logits to bf16 like the reference's baddbmm; upcast_attention (f32 logits) is left
        # to the framework's attention
        if ((query.dtype is torch.float16 or (query.dtype is torch.bfloat16 and round_logits))
                and query.dim() == 3 and key.dim() == 3 and key.shape[1] == self.tokens
                and d * heads == query.shape[2] and key.shape[2] == query.shape[2] and query.shape[0] == key.shape[0]
                and d % 8 == 0 and 8 <= d <= 160 and query.shape[1] % 8 == 0):
            self._ensure_ctx(query.dtype)
            if self.acc_dtype in (query.dtype, torch.float32):
                b, hw, c = query.shape
                qk = nat.QKDesc(in_dtype=_DTYPE_CODE[query.dtype], batch=b, heads=heads, hw=hw, tokens=self.tokens, head_dim=d,
                                round_logits=1 if round_logits else 0, scale=float(scale),
                                q_stride_b=hw * c, q_stride_h=d, q_stride_p=c,
                                k_stride_b=self.tokens * c, k_stride_h=d, k_stride_t=c)
                desc = nat.AttendDesc(qk=qk, v_stride_b=self.tokens * c, v_stride_h=d, v_stride_t=c,
                                      o_stride_b=hw * c, o_stride_h=d, o_stride_p=c)
        entry = CallShape(query, key, heads, scale, round_logits, 0, desc)
        self._att_cache[layer] = entry
        if self._fast is not None:
            self._fast.set_attend_cache(layer, query, key, int(heads), float(scale), bool(round_logits), entry.desc_addr)
        return entry

    def _launch_stream(self):
        """The stream a deferred launch goes to, ordered after the producers of the recorded Q / K: normally the
        current stream IS the stream they were produced on; when the maps are read from another stream (generation
        inside ``torch.cuda.stream(s)``, ``compute_global_heat_map`` outside), the launch waits for everything queued
        on the recording stream so far.  Returns ``(stream handle, recording stream or None)``."""
        cur = self._current_stream()
        rec = self._rec_stream
        if rec is None or rec.cuda_stream == cur.cuda_stream:
            return cur.cuda_stream, None
        cur.wait_stream(rec)
        return cur.cuda_stream, rec

    def flush(self, _before_launch=None) -> bool:
        """Run every recorded (deferred) tap; the held Q/K references are dropped afterwards
        (stream order keeps their memory valid until the kernel has consumed it).  ``_before_launch(stream)`` is called between
        handing the recorded calls to the library and the launch (``global_heat_map`` announces its output there, so that the
        launch's table-upload kernel clears it).  Returns whether anything was launched."""
        if self._fast is not None:
            n, la, qa, ka, da = self._fast.buffers()
            if self.ctx is None or n == 0:
                return False
            try:
                stream, rec_stream = self._launch_stream()
                nat.check(self.lib.daam_tap_qk_enqueue_many(self.ctx, n, la, qa, ka, da))
                self._announce_then_launch(_before_launch, stream)
                if rec_stream is not None:
                    # the Q / K blocks return to the recording stream's allocator pool: not before the tap has read them
                    rec_stream.wait_stream(self._current_stream())
                self._set_window(self.defer_steps)
            finally:
                self._drop_recorded()
            return True
        rec = self._rec
        n = len(rec)
        if self.ctx is None or n == 0:
            return False
        if self._check_versions:
            for layer, q, k, _d, qv, kv in rec:
                if q._version != qv or k._version != kv:
                    self._drop_recorded()
                    raise RuntimeError(f'daam_amd: the query / key of layer {layer} was modified in place between its '
                                       'attention call and the deferred tap launch (use defer_steps=0 for such a pipeline)')
            rec = [r[:4] for r in rec]
        lay_t, q_t, k_t, d_t = zip(*rec)                       # one C-level pass
        layers = np.array(lay_t, dtype=np.int32)
        qp = np.array([t.data_ptr() for t in q_t], dtype=np.uint64)
        kp = np.array([t.data_ptr() for t in k_t], dtype=np.uint64)
        dp = np.array(d_t, dtype=np.uint64)
        try:
            stream, rec_stream = self._launch_stream()
            nat.check(self.lib.daam_tap_qk_enqueue_many(self.ctx, n, layers.ctypes.data, qp.ctypes.data, kp.ctypes.data,
                                                        dp.ctypes.data))
            self._announce_then_launch(_before_launch, stream)
            if rec_stream is not None:
                rec_stream.wait_stream(self._current_stream())
            self._window = self.defer_steps
        finally:
            self._drop_recorded()
        return True

    def _announce_then_launch(self, before_launch, stream) -> None:
        """The recorded calls are in the library's hands: whatever ``before_launch`` does, the launch that consumes (and drops) them
        must follow -- the Python references to their Q / K are released right after, and entries left pending would be read
        from freed memory by the next launch.  A failing announcement is re-raised after the launch."""
        failed = None
        if before_launch is not None:
            try:
                before_launch(stream)
            except BaseException as e:                         # noqa: BLE001 -- re-raised below
                failed = e
        rc = self.lib.daam_tap_flush(self.ctx, stream)
        if failed is not None:
            raise failed
        nat.check(rc)

    def _drop_recorded(self) -> None:
        self._rec.clear()
        self._held = 0
        self._cnt[:] = [0] * self.n_layers              # in place: tap_qk holds a reference across flush()
        if self._fast is not None:
            self._fast.drop()

    def tap_probs(self, layer: int, probs: torch.Tensor, factor: int) -> None:
        """``probs`` [B*H, hw, tokens] as returned by ``get_attention_scores`` (trace.py:276)."""
        self._require_device(probs)
        self._ensure_ctx(probs.dtype)
        self.flush()
        probs = probs if probs.is_contiguous() else probs.contiguous()
        bh, hw, tokens = probs.shape
        self._ensure_layer(layer, bh - bh // 2, int(math.sqrt(hw)), factor)
        nat.check(self.lib.daam_tap_probs(self.ctx, layer, probs.data_ptr(),
                                          _DTYPE_CODE[probs.dtype],
                                          bh, hw, tokens, self.stream))
        self._touch(layer)

    def add_map(self, factor: int, layer: int, head: int, heat_map: torch.Tensor) -> None:
        """``RawHeatMapCollection.update`` called by hand (heatmap.py:153-156): rare, done with a
        torch add on the layer's buffer."""
        self._require_device(heat_map)
        self._ensure_ctx(heat_map.dtype)
        self.flush()
        t, h, w = heat_map.shape
        if layer not in self.layer_info:
            raise RuntimeError('daam_amd: update() on a layer that was never tapped is not supported')
        # clear() is lazy on the device side: the library zeroes the buffer now if it still owes that (on this
        # stream) and learns that the layer holds sums again (later taps add to them, the next reset clears them)
        nat.check(self.lib.daam_layer_touch(self.ctx, layer, self.stream))
        self.acc[layer][head] += heat_map.to(self.acc_dtype)
        self._touch(layer)

    # ---- views -----------------------------------------------------------------------------------
    def keys(self) -> List[Key]:
        out: List[Key] = []
        for layer in self.touched:
            factor, heads, _ = self.layer_info[layer]
            out += [(factor, layer, h) for h in range(heads)]
        return out

    def items(self) -> Iterator[Tuple[Key, torch.Tensor]]:
        """``((factor, layer, head), running sum [tokens, h, w])`` in first-update order.  The tensors are VIEWS of
        the live sum buffers (no 221 MB copy per iteration); they stay valid and unchanged after ``clear()`` / the
        end of the trace -- like the reference's tensors -- because an engine whose buffers were handed out starts its
        next generation on fresh buffers and does not park them for reuse.  Taps of the SAME generation that follow
        the iteration do show up in them."""
        self.flush()
        self._views_out = True
        for layer in list(self.touched):
            factor, heads, _ = self.layer_info[layer]
            buf = self.acc[layer]
            for h in range(heads):
                yield (factor, layer, h), buf[h]

    # ---- finalize ---------------------------------------------------------------------------------
    def global_heat_map(self, factors: Optional[Sequence[int]] = None, head_idx: Optional[int] = None,
                        layer_idx: Optional[int] = None, n_rows: Optional[int] = None) -> torch.Tensor:
        """trace.py:103-126: returns ``[n_rows, x, x]`` fp32 on the device.  ``n_rows`` (default: every token row) is the
        crop of trace.py:127 applied BEFORE the work: the planes of the token rows nobody reads are neither fetched nor
        written (``daam_finalize``'s ``n_rows``, ABI v6)."""
        rows = self.tokens if n_rows is None else max(1, min(int(n_rows), self.tokens))
        fset = {0, 1, 2, 4, 8, 16, 32, 64} if factors is None else set(factors)
        if self.ctx is None or not self.touched:
            raise LookupError('no heat maps')
        # key mask in the library's key order (configured layers by index, heads inside); cached per
        # selection -- building it costs more host time than the finalize kernels take on the device
        sel = (tuple(sorted(fset)), head_idx, layer_idx, tuple(self.touched), len(self.layer_info))
        cached = self._mask_cache.get(sel)
        if cached is None:
            total = ctypes.c_int()
            nat.check(self.lib.daam_key_offset(self.ctx, 0, None, ctypes.byref(total)))
            mask = (ctypes.c_uint8 * total.value)()
            n = 0
            for layer in self.touched:
                factor, heads, _ = self.layer_info[layer]
                if factor not in fset or (layer_idx is not None and layer_idx != layer):
                    continue
                off = ctypes.c_int()
                nat.check(self.lib.daam_key_offset(self.ctx, layer, ctypes.byref(off), None))
                for h in range(heads):
                    if head_idx is None or head_idx == h:
                        mask[off.value + h] = 1
                        n += 1
            if len(self._mask_cache) > 64:
                self._mask_cache.clear()
            cached = self._mask_cache[sel] = (mask, n)
        mask, n = cached
        if n == 0:
            self.flush()
            raise LookupError('no heat maps')
        out = torch.empty(rows, self.out_side, self.out_side, dtype=torch.float32, device=self.device)
        # the output is announced BEFORE the deferred taps go out: the launch's table-upload kernel clears it and the key tables
        # stay on the device between generations, so the finalize call below is its class kernel(s) only (daam_finalize_prepare)
        optr = out.data_ptr()

        def announce(stream):
            nat.check(self.lib.daam_finalize_prepare(self.ctx, mask, rows, optr, stream))
        if not self.flush(_before_launch=announce):
            announce(self.stream)
        nat.check(self.lib.daam_finalize(self.ctx, mask, rows, optr, self.stream))
        return out

    def normalize_(self, maps: torch.Tensor) -> torch.Tensor:
        """trace.py:129-130, in place on ``maps`` [n_rows, x, x] (contiguous fp32)."""
        nat.check(self.lib.daam_epilogue_normalize(maps.data_ptr(), maps.shape[0], maps.shape[-1], self.stream))
        return maps


def word_heat_map(maps: torch.Tensor, idxs: Sequence[int]) -> torch.Tensor:
    """heatmap.py:121-123: mean of the planes ``idxs`` of ``maps`` [rows, s, s] -> [s, s]."""
    lib = nat.load()
    _check_maps(maps)
    side = maps.shape[-1]
    idx = (ctypes.c_int32 * len(idxs))(*[int(i) for i in idxs])
    for i in idxs:
        if not 0 <= int(i) < maps.shape[0]:
            raise IndexError(f'index {i} is out of bounds for dimension 0 with size {maps.shape[0]}')
    word = torch.empty(side, side, dtype=torch.float32, device=maps.device)
    ws = torch.empty(2, dtype=torch.float32, device=maps.device)
    nat.check(lib.daam_word_heat_map(maps.data_ptr(), side, idx, len(idxs), word.data_ptr(), None, 0, 0, 1, 0.0,
                                     ws.data_ptr(), torch.cuda.current_stream(maps.device).cuda_stream))
    return word


def expand_word_map(word: torch.Tensor, out_h: int, out_w: int, absolute: bool = False,
                    threshold: Optional[float] = None) -> torch.Tensor:
    """heatmap.py:77-93 up to (not including) the ``.cpu()``: bicubic to ``out_h x out_w``,
    min-max normalise unless ``absolute``, optional threshold."""
    lib = nat.load()
    if word.device.type != 'cuda' or word.dtype != torch.float32 or word.dim() != 2 or word.shape[0] != word.shape[1]:
        raise RuntimeError('daam_amd: expand_as needs a square fp32 word map on the HIP device')
    word = word.contiguous()
    side = word.shape[-1]
    out = torch.empty(out_h, out_w, dtype=torch.float32, device=word.device)
    tmp = torch.empty(side, side, dtype=torch.float32, device=word.device)
    ws = torch.empty(2, dtype=torch.float32, device=word.device)
    idx = (ctypes.c_int32 * 1)(0)
    nat.check(lib.daam_word_heat_map(word.data_ptr(), side, idx, 1, tmp.data_ptr(), out.data_ptr(), out_h, out_w,
                                     1 if absolute else 0, float(threshold) if threshold else 0.0, ws.data_ptr(),
                                     torch.cuda.current_stream(word.device).cuda_stream))
    return out


def _check_maps(maps: torch.Tensor) -> None:
    if maps.device.type != 'cuda':
        raise RuntimeError('daam_amd: heat maps must live on the HIP device (no CPU fallback)')
    if maps.dtype != torch.float32 or not maps.is_contiguous() or maps.dim() != 3 or maps.shape[1] != maps.shape[2]:
        raise RuntimeError('daam_amd: heat maps must be a contiguous fp32 [rows, s, s] tensor')

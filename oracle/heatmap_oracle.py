"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by ``daam_amd`` (the product path).

A CPU (numpy) restatement of the DAAM heat-map extraction path of castorini/daam v0.2.0.
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this module, and only as the checker.

Parity status: **pinned against the reference itself** -- the reference's own code
(``/root/reference/daam``) is executed unmodified in this container through the stubs in
``oracle/fake_diffusers.py`` and its outputs are committed as ``tests/golden/*.npz`` by
``oracle/make_golden.py``; ``tests/test_oracle_golden.py`` checks this restatement against
those vectors.  The reference ships no tests / golden vectors of its own.

Each function cites the reference lines it follows (paths relative to /root/reference).

Numeric modes (``pipe_dtype``):
  * ``float32``  -- literal fp32 pipeline (what the reference does on CPU, config C1).
  * ``float16``  -- literal fp16 pipeline as PyTorch-ROCm executes it: logits rounded to
                    fp16 (``baddbmm`` output dtype), softmax internally fp32, probabilities
                    rounded to fp16, running sums kept in fp16 (``heatmap.py:150,156`` --
                    ``add`` has no autocast rule), bicubic / clamp / mean in fp32
                    (autocast FP32 policy, ``trace.py:111``).
  * ``'bfloat16'`` -- literal bf16 pipeline, same rounding points as fp16 (bf16 logits,
                    probabilities and running sums).  numpy has no bfloat16: such arrays are carried
                    as float32 holding bf16-representable values (``round_bf16``); an f32 add of two
                    of them followed by one rounding IS the correctly rounded bf16 add.
  * ``float64``  -- ground truth (no intermediate rounding) for accuracy reporting.
``acc_dtype`` may override the running-sum type (the ``f32`` accuracy mode of the kernels).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np

Key = Tuple[int, int, int]          # (factor, layer, head) -- heatmap.py:145

_BICUBIC_A = -0.75                  # torch upsample_bicubic2d constant

BF16 = 'bfloat16'


def is_bf16(dt) -> bool:
    return isinstance(dt, str) and dt == BF16


def round_bf16(x) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32).reshape(np.shape(x))


def _cast(x: np.ndarray, dt) -> np.ndarray:
    return round_bf16(x) if is_bf16(dt) else x.astype(dt)


# --------------------------------------------------------------------------------------
# K1 / K2: diffusers Attention.get_attention_scores as called at trace.py:276
# --------------------------------------------------------------------------------------
def attention_probs(q: np.ndarray, k: np.ndarray, scale: float, pipe_dtype=np.float32,
                    upcast_attention: bool = False, mask: Optional[np.ndarray] = None) -> np.ndarray:
    """``softmax(scale * q k^T, -1)`` with the reference pipeline's rounding points.

    q ``[BH, hw, d]``, k ``[BH, T, d]`` -> probs ``[BH, hw, T]`` in ``pipe_dtype``.
    (diffusers 0.21.2 ``get_attention_scores``: ``baddbmm(beta=0, alpha=scale)`` ->
    ``softmax(dim=-1)`` -> ``.to(dtype)``; SURVEY Appendix A.)  ``upcast_attention``: q, k are up-cast first, so
    the logits stay f32 (no rounding to the pipeline dtype).  ``upcast_softmax`` changes nothing numerically: the
    softmax of fp16 logits already runs in f32 internally.  ``mask``: additive bias ``[BH, 1 | hw, T]``
    (``baddbmm(mask, q, k^T, beta=1)``: added in f32 before the rounding).
    """
    if not is_bf16(pipe_dtype):
        pipe_dtype = np.dtype(pipe_dtype)
    if pipe_dtype == np.float64:
        logits = np.einsum('bpd,btd->bpt', q.astype(np.float64), k.astype(np.float64)) * float(scale)
        m = logits.max(-1, keepdims=True)
        e = np.exp(logits - m)
        return e / e.sum(-1, keepdims=True)
    # fp32 accumulate, alpha applied in fp32, result rounded to the pipe dtype
    acc = np.matmul(q.astype(np.float32), np.swapaxes(k.astype(np.float32), -1, -2))
    acc = acc * np.float32(scale)
    if mask is not None:
        acc = acc + mask.astype(np.float32)
    logits = acc if upcast_attention else _cast(acc, pipe_dtype)
    x = logits.astype(np.float32)
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m, dtype=np.float32)
    p = e / e.sum(-1, keepdims=True, dtype=np.float32)
    return _cast(p, pipe_dtype)


def attention_output(q: np.ndarray, k: np.ndarray, v: np.ndarray, scale: float, pipe_dtype=np.float32,
                     upcast_attention: bool = False, mask: Optional[np.ndarray] = None) -> np.ndarray:
    """What the reference's processor computes between the projections and ``to_out``: ``torch.bmm(attention_probs,
    value)`` (trace.py:296) on the probabilities of ``get_attention_scores`` (trace.py:276), in the pipeline dtype.

    q ``[BH, hw, d]``, k / v ``[BH, T, d]`` -> ``[BH, hw, d]`` in ``pipe_dtype``: the probabilities are rounded to the
    pipeline dtype (``attention_probs``), the product with ``value`` is accumulated wide and rounded ONCE (a GEMM with
    f32 accumulation; summed in f64 here, which differs from any f32 order by less than the final rounding)."""
    probs = attention_probs(q, k, scale, pipe_dtype, upcast_attention, mask)
    wide = np.matmul(probs.astype(np.float64), v.astype(np.float64))
    if not is_bf16(pipe_dtype) and np.dtype(pipe_dtype) == np.float64:
        return wide
    return _cast(wide.astype(np.float32), pipe_dtype)


def batch_to_head_dim(t: np.ndarray, heads: int) -> np.ndarray:
    """diffusers ``Attention.batch_to_head_dim`` (trace.py:297): ``[B*H, S, d] -> [B, S, H*d]``."""
    bh, s, d = t.shape
    return t.reshape(bh // heads, heads, s, d).transpose(0, 2, 1, 3).reshape(bh // heads, s, heads * d)


# --------------------------------------------------------------------------------------
# K3: UNetCrossAttentionHooker._unravel_attn  (trace.py:219-244)
# --------------------------------------------------------------------------------------
def unravel(probs: np.ndarray) -> np.ndarray:
    """``[BH, hw, T] -> [BH - BH//2, T, h, w]``: keep the second half of the batch*heads
    dim (``map_[map_.size(0) // 2:]``, trace.py:240 = the conditional prompt under CFG),
    reshape pixels to a square (trace.py:233) and move tokens in front of pixels."""
    bh, hw, t = probs.shape
    side = int(math.sqrt(hw))
    kept = probs[bh // 2:]
    return np.ascontiguousarray(kept.transpose(0, 2, 1)).reshape(bh - bh // 2, t, side, side)


def layer_factor(latent_hw: int, hw: int) -> int:
    """trace.py:285."""
    return int(math.sqrt(latent_hw // hw))


def latent_hw_for(sample_size: int, vae_scale_factor: int) -> int:
    """trace.py:32-33."""
    h = sample_size * vae_scale_factor
    return 4096 if h in (512, 1024) else 9216


# --------------------------------------------------------------------------------------
# K4: RawHeatMapCollection (heatmap.py:148-172) + the gate at trace.py:285-294
# --------------------------------------------------------------------------------------
class RawMaps:
    """Insertion-ordered ``(factor, layer, head) -> running sum [T, h, w]``."""

    def __init__(self, acc_dtype=np.float32):
        self.acc_dtype = acc_dtype if is_bf16(acc_dtype) else np.dtype(acc_dtype)
        self.maps: Dict[Key, np.ndarray] = {}

    def update(self, factor: int, layer: int, head: int, heat_map: np.ndarray):
        key = (factor, layer, head)
        prev = self.maps.get(key)
        add = _cast(heat_map, self.acc_dtype)
        # out-of-place ``acc = acc + map`` in the accumulator dtype (heatmap.py:156).
        # numpy's float16 add computes in fp32 and rounds once (RNE) == torch's fp16 add.
        if prev is None:
            self.maps[key] = add.copy()
        else:
            self.maps[key] = round_bf16(prev + add) if is_bf16(self.acc_dtype) else prev + add

    def clear(self):
        self.maps.clear()

    def __iter__(self):
        return iter(self.maps.items())

    def __len__(self):
        return len(self.maps)


def tap(raw: RawMaps, layer_idx: int, q: np.ndarray, k: np.ndarray, scale: float,
        latent_hw: int, pipe_dtype=np.float32, context_size: int = 77,
        probs: Optional[np.ndarray] = None, upcast_attention: bool = False) -> Optional[np.ndarray]:
    """One hooked cross-attention call (trace.py:276-294): scores -> gate -> unravel ->
    per-head update.  Returns the probabilities (the reference needs them for ``bmm``)."""
    if probs is None:
        probs = attention_probs(q, k, scale, pipe_dtype, upcast_attention)
    factor = layer_factor(latent_hw, probs.shape[1])
    if probs.shape[-1] == context_size and factor != 8:          # trace.py:289
        maps = unravel(probs)
        for head_idx in range(maps.shape[0]):                    # trace.py:293-294
            raw.update(factor, layer_idx, head_idx, maps[head_idx])
    return probs


# --------------------------------------------------------------------------------------
# K6: F.interpolate(mode='bicubic', align_corners=False, antialias=False)  (trace.py:116)
# --------------------------------------------------------------------------------------
def bicubic_taps(in_size: int, out_size: int, dtype=np.float32):
    """Tap indices ``[out,4]`` (border-clamped) and weights ``[out,4]`` (A = -0.75).

    torch: ``src = scale * (dst + 0.5) - 0.5`` with ``scale = in/out`` (computed in the
    accumulate type), ``f = floor(src)``, ``t = src - f``, taps ``f-1 .. f+2`` clamped to
    ``[0, in-1]``;  coefficients per ``get_cubic_upsample_coefficients``.
    """
    dt = np.dtype(dtype)
    scale = dt.type(in_size) / dt.type(out_size)
    dst = np.arange(out_size, dtype=dt)
    src = scale * (dst + dt.type(0.5)) - dt.type(0.5)
    f = np.floor(src)
    t = (src - f).astype(dt)
    a = dt.type(_BICUBIC_A)

    def conv1(x):   # |x| <= 1
        return ((a + 2) * x - (a + 3)) * x * x + 1

    def conv2(x):   # 1 < |x| < 2
        return ((a * x - 5 * a) * x + 8 * a) * x - 4 * a

    w = np.stack([conv2(t + 1), conv1(t), conv1(1 - t), conv2(2 - t)], axis=1).astype(dt)
    idx = f.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :]
    idx = np.clip(idx, 0, in_size - 1)
    return idx, w


def bicubic_resize(planes: np.ndarray, out_size: int, dtype=np.float32) -> np.ndarray:
    """``[..., h, w] -> [..., out, out]``; identical sizes are a bit-exact copy (torch
    short-circuits that case)."""
    h, w = planes.shape[-2:]
    x = planes.astype(dtype)
    if h == out_size and w == out_size:
        return x.copy()
    ix, wx = bicubic_taps(w, out_size, dtype)
    iy, wy = bicubic_taps(h, out_size, dtype)
    # torch interpolates along x on the four source rows, then along y
    rows = np.zeros(x.shape[:-1] + (out_size,), dtype=dtype)
    for b in range(4):
        rows += x[..., :, ix[:, b]] * wx[:, b]
    out = np.zeros(x.shape[:-2] + (out_size, out_size), dtype=dtype)
    for a_ in range(4):
        out += rows[..., iy[:, a_], :] * wy[:, a_][:, None]
    return out


# --------------------------------------------------------------------------------------
# K6-K8: DiffusionHeatMapHooker.compute_global_heat_map  (trace.py:83-132)
# --------------------------------------------------------------------------------------
def global_heat_map(raw: Iterable, latent_hw: int, n_rows: Optional[int] = None,
                    factors: Optional[Sequence[int]] = None, head_idx: Optional[int] = None,
                    layer_idx: Optional[int] = None, normalize: bool = False,
                    dtype=np.float32) -> np.ndarray:
    """Per selected key: bicubic to ``x = int(sqrt(latent_hw))`` -> ``clamp(min=0)``
    (trace.py:114-116); mean over keys (trace.py:119,126); crop to ``n_rows`` =
    ``len(tokenize(prompt)) + 2`` (trace.py:127); optional per-pixel normalisation over the
    content tokens ``1..-2`` (trace.py:129-130).  Raises RuntimeError like trace.py:120-124."""
    fset = {0, 1, 2, 4, 8, 16, 32, 64} if factors is None else set(factors)   # trace.py:103-106
    x = int(np.sqrt(latent_hw))                                                # trace.py:109
    total, n = None, 0
    for (factor, layer, head), hm in raw:
        if factor in fset and (head_idx is None or head_idx == head) and \
                (layer_idx is None or layer_idx == layer):                     # trace.py:113
            up = np.maximum(bicubic_resize(hm, x, dtype), 0)
            total = up.astype(dtype) if total is None else total + up
            n += 1
    if n == 0:
        if head_idx is not None or layer_idx is not None:
            raise RuntimeError('No heat maps found for the given parameters.')
        raise RuntimeError('No heat maps found. Did you forget to call `with trace(...)` during generation?')
    maps = total * np.dtype(dtype).type(1.0 / n)
    if n_rows is not None:
        maps = maps[:n_rows]
    if normalize:
        maps = maps / (maps[1:-1].sum(0, keepdims=True) + np.dtype(dtype).type(1e-6))
    return maps


# --------------------------------------------------------------------------------------
# K9 (next row f1): GlobalHeatMap.compute_word_heat_map + WordHeatMap.expand_as
# --------------------------------------------------------------------------------------
def token_merge_indices(tokens: Sequence[str], word_tokens: Sequence[str], word: str,
                        word_idx: Optional[int] = None, offset_idx: int = 0):
    """utils.py:73-91 on pre-tokenised input (lower-casing / ``</w>`` stripping done here)."""
    toks = [t.replace('</w>', '') for t in tokens]
    if word_idx is not None:
        return [word_idx + 1], word_idx
    search = [t.replace('</w>', '') for t in word_tokens]
    merge = []
    for start in range(len(toks)):
        if toks[start:start + len(search)] == search:
            merge += [start + offset_idx + i for i in range(len(search))]
    if not merge:
        raise ValueError(f'Search word {word} not found in prompt!')
    return [m + 1 for m in merge], word_idx          # +1: SOS offset (utils.py:91)


def word_heat_map(global_maps: np.ndarray, merge_idxs: Sequence[int]) -> np.ndarray:
    """heatmap.py:121-123."""
    return global_maps[list(merge_idxs)].mean(0)


def expand_as(word_map: np.ndarray, out_size: int, absolute: bool = False,
              threshold: Optional[float] = None) -> np.ndarray:
    """heatmap.py:77-93: bicubic to the image size, min-max normalise, optional threshold."""
    im = bicubic_resize(word_map.astype(np.float32), out_size, np.float32)
    if not absolute:
        im = (im - im.min()) / (im.max() - im.min() + np.float32(1e-8))
    if threshold:
        im = (im > threshold).astype(np.float32)
    return im


# --------------------------------------------------------------------------------------
# f4: evaluate.compute_iou / compute_ioa  (evaluate.py:14-35)
# --------------------------------------------------------------------------------------
def _resize_binarise(a: np.ndarray, b_shape) -> np.ndarray:
    """evaluate.py:15-18: only when the HEIGHTS differ; bicubic to b's (h, w), then a < 1 -> 0, a >= 1 -> 1."""
    a = a.astype(np.float32)
    if a.shape[0] == b_shape[0]:
        return a
    ih, iw = a.shape
    oh, ow = b_shape
    ix, wx = bicubic_taps(iw, ow)
    iy, wy = bicubic_taps(ih, oh)
    rows = np.zeros((ih, ow), dtype=np.float32)
    for t in range(4):
        rows += a[:, ix[:, t]] * wx[:, t]
    out = np.zeros((oh, ow), dtype=np.float32)
    for t in range(4):
        out += rows[iy[:, t], :] * wy[:, t][:, None]
    # two masked assignments in the reference: a NaN satisfies neither and stays a NaN
    return np.where(out < 1, np.float32(0), np.where(out >= 1, np.float32(1), out)).astype(np.float32)


def mask_overlap(a: np.ndarray, b: np.ndarray):
    a = _resize_binarise(a, b.shape)
    b = b.astype(np.float32)
    return np.float32((a * b).sum(dtype=np.float32)), np.float32(a.sum(dtype=np.float32)), np.float32(b.sum(dtype=np.float32))


def compute_iou(a: np.ndarray, b: np.ndarray) -> float:
    inter, sa, sb = mask_overlap(a, b)
    return float(inter / (sa + sb - inter + np.float32(1e-8)))


def compute_ioa(a: np.ndarray, b: np.ndarray) -> float:
    inter, sa, _ = mask_overlap(a, b)
    return float(inter / (sa + np.float32(1e-8)))


# --------------------------------------------------------------------------------------
# a1: UNetCrossAttentionLocator.locate  (hook.py:95-127)
# --------------------------------------------------------------------------------------
def locate(unet, restrict=None, locate_middle_block: bool = False):
    """Ordered ``attn2`` modules: up blocks, then down blocks, then (optionally) mid;
    only blocks whose class name contains ``'CrossAttn'`` (hook.py:115); ``restrict``
    filters positions inside each block (hook.py:122); names restart per block (hook.py:123).
    Returns ``(modules, names)``; ``layer_idx`` is the list position (trace.py:45,50)."""
    groups = [(b, 'up') for b in unet.up_blocks] + [(b, 'down') for b in unet.down_blocks]
    if locate_middle_block:
        groups.append((unet.mid_block, 'mid'))
    modules, names = [], []
    for block, tag in groups:
        if 'CrossAttn' not in type(block).__name__:
            continue
        found = [tb.attn2 for tr in block.attentions for tb in tr.transformer_blocks]
        kept = [m for i, m in enumerate(found) if restrict is None or i in restrict]
        modules += kept
        names += [f'{tag}-attn-{i}' for i in range(len(kept)) if restrict is None or i in restrict]
    return modules, names


def replay_generation(pipe, steps: int, pipe_dtype, acc_dtype=None, restrict=None,
                      locate_middle_block: bool = False, outputs: Optional[list] = None) -> "RawMaps":
    """Drive the oracle over the same synthetic inputs a fake pipeline feeds its UNet
    (``oracle/fake_diffusers.py``): per step, every cross-attention in execution order;
    hooked ones (per ``locate``) are tapped with ``layer_idx`` = locator position.
    ``outputs``: a list that receives what every processor call of the LAST step returns (trace.py:296-304:
    ``to_out(batch_to_head_dim(bmm(probs, value)))``), execution order, hooked or not."""
    import torch
    np_dtype = {torch.float16: np.float16, torch.float32: np.float32, torch.bfloat16: BF16,
                torch.float64: np.float64}.get(pipe_dtype, pipe_dtype)
    as_np = (lambda t: t.detach().float().cpu().numpy()) if is_bf16(np_dtype) else (lambda t: t.detach().cpu().numpy())
    raw = RawMaps(np_dtype if acc_dtype is None else acc_dtype)
    modules, _ = locate(pipe.unet, restrict, locate_middle_block)
    index_of = {id(m): i for i, m in enumerate(modules)}
    lat = latent_hw_for(pipe.unet.config.sample_size, pipe.vae_scale_factor)
    order = pipe.unet.execution_order()
    for step in range(steps):
        for i, spec in enumerate(order):
            li = index_of.get(id(spec.module))
            want_out = outputs is not None and step == steps - 1
            if li is None and not want_out:
                continue
            a = spec.module
            q = as_np(a.head_to_batch_dim(a.to_q(pipe.hidden_states(i, spec, step))))
            k = as_np(a.head_to_batch_dim(a.to_k(pipe.context(i, spec))))
            upcast = getattr(a, 'upcast_attention', False)
            if li is not None:
                tap(raw, li, q, k, a.scale, lat, np_dtype, upcast_attention=upcast)
            if want_out:
                v = as_np(a.head_to_batch_dim(a.to_v(pipe.context(i, spec))))
                mixed = batch_to_head_dim(attention_output(q, k, v, a.scale, np_dtype, upcast_attention=upcast), a.heads)
                w, b = as_np(a.to_out[0].weight.detach()), a.to_out[0].bias
                proj = np.matmul(mixed.astype(np.float64), w.astype(np.float64).T)
                if b is not None:
                    proj = proj + as_np(b.detach()).astype(np.float64)
                outputs.append(proj if np_dtype == np.float64 else _cast(proj.astype(np.float32), np_dtype))
    return raw

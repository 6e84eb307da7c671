"""TEST / BENCH INFRASTRUCTURE -- a torch restatement ("port") of the reference's hook path, used
(a) by ``bench.py`` as the timed CPU baseline (``cpu_baseline.kind = "port"``) and as the
"reference hooks in PyTorch eager on the MI355X" comparison, both of which must run on the GPU
box where ``/root/reference`` does not exist, and (b) by ``tests/test_oracle_golden.py``, which
pins it to the golden vectors of the unmodified reference.  Never imported by ``daam_amd``.

It performs the same torch op sequence as the reference (that is the point: the op count IS
the overhead being measured):
  * ``unravel``  = UNetCrossAttentionHooker._unravel_attn  (daam/trace.py:219-244)
  * ``RawMaps``  = RawHeatMapCollection                     (daam/heatmap.py:148-172)
  * ``tap``      = the DAAM-specific part of __call__       (daam/trace.py:285-294)
  * ``global_heat_map`` = compute_global_heat_map           (daam/trace.py:103-130)
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Optional, Sequence

import torch
import torch.nn.functional as F


def _auto_autocast(*args, **kwargs):
    """The reference's ``auto_autocast`` (daam/utils.py:32-36), which wraps the unravel loop (trace.py:237), every
    ``update`` (heatmap.py:154) and ``compute_global_heat_map`` (trace.py:111): entering it 1100+ times per SDXL denoising
    step is part of the host cost being timed (tools/port_vs_reference_cpu.py: without it the port ran 24 % faster than the
    reference).  It changes no value here: none of the wrapped ops of the per-step path is an autocast op, and the bicubic
    input is up-cast explicitly below."""
    if not torch.cuda.is_available():
        kwargs['enabled'] = False
    return torch.cuda.amp.autocast(*args, **kwargs)


@torch.no_grad()
def unravel(x: torch.Tensor) -> torch.Tensor:
    """``[BH, hw, tokens] -> [kept, tokens, h, w]`` by the reference's op sequence: permute,
    one view + slice per token (77 Python iterations), stack, permute, contiguous."""
    side = int(math.sqrt(x.size(1)))
    planes = []
    x = x.permute(2, 0, 1)                                      # trace.py:235
    with _auto_autocast(dtype=torch.float32):                   # trace.py:237
        for per_token in x:                                     # trace.py:238
            per_token = per_token.view(per_token.size(0), side, side)
            planes.append(per_token[per_token.size(0) // 2:])  # conditional half, trace.py:240
    return torch.stack(planes, 0).permute(1, 0, 2, 3).contiguous()


class RawMaps:
    def __init__(self):
        self.maps = OrderedDict()

    def update(self, factor: int, layer: int, head: int, heat_map: torch.Tensor):
        with _auto_autocast(dtype=torch.float32):                           # heatmap.py:154
            key = (factor, layer, head)
            prev = self.maps.get(key)
            self.maps[key] = heat_map + 0.0 if prev is None else prev + heat_map   # out-of-place add, heatmap.py:156

    def clear(self):
        self.maps.clear()

    def __iter__(self):
        return iter(self.maps.items())

    def __len__(self):
        return len(self.maps)


@torch.no_grad()
def tap(raw: RawMaps, layer_idx: int, probs: torch.Tensor, latent_hw: int, context_size: int = 77) -> None:
    factor = int(math.sqrt(latent_hw // probs.shape[1]))                    # trace.py:285
    if probs.shape[-1] == context_size and factor != 8:                     # trace.py:289
        maps = unravel(probs)
        for head_idx, heat_map in enumerate(maps):                          # trace.py:293-294
            raw.update(factor, layer_idx, head_idx, heat_map)


class ReferenceProcessor:
    """The reference's attention processor (``UNetCrossAttentionHooker.__call__``, trace.py:252-304) as a plain
    torch op sequence: mask preparation with the QUERY length, optional ``norm_cross``, materialised probabilities from
    ``attn.get_attention_scores``, the DAAM tap (``tap`` above), ``bmm`` and the output projection.  ``raw=None``
    skips the tap (then it is what a stock materialising processor computes)."""

    def __init__(self, raw: Optional['RawMaps'] = None, layer_idx: int = 0, latent_hw: int = 4096):
        self.raw, self.layer_idx, self.latent_hw = raw, layer_idx, latent_hw

    @torch.no_grad()
    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **_):
        batch_size, sequence_length, _c = hidden_states.shape
        attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)   # trace.py:259-260
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        elif attn.norm_cross is not None:
            encoder_hidden_states = attn.norm_cross(encoder_hidden_states)                         # trace.py:264-267
        key = attn.head_to_batch_dim(attn.to_k(encoder_hidden_states))
        value = attn.head_to_batch_dim(attn.to_v(encoder_hidden_states))
        query = attn.head_to_batch_dim(query)
        probs = attn.get_attention_scores(query, key, attention_mask)                              # trace.py:276
        if self.raw is not None:
            tap(self.raw, self.layer_idx, probs, self.latent_hw)                                    # trace.py:285-294
        out = attn.batch_to_head_dim(torch.bmm(probs, value))                                      # trace.py:296-297
        return attn.to_out[1](attn.to_out[0](out))                                                 # trace.py:300-302


@torch.no_grad()
def attention_probs(q: torch.Tensor, k: torch.Tensor, scale: float) -> torch.Tensor:
    """diffusers 0.21.2 get_attention_scores (no mask, no upcast) as called at trace.py:276."""
    base = torch.empty(q.shape[0], q.shape[1], k.shape[1], dtype=q.dtype, device=q.device)
    scores = torch.baddbmm(base, q, k.transpose(-1, -2), beta=0, alpha=scale)
    return scores.softmax(dim=-1).to(q.dtype)


@torch.no_grad()
def global_heat_map(raw, latent_hw: int, n_rows: Optional[int] = None, factors: Optional[Sequence[int]] = None,
                    head_idx: Optional[int] = None, layer_idx: Optional[int] = None,
                    normalize: bool = False) -> torch.Tensor:
    fset = {0, 1, 2, 4, 8, 16, 32, 64} if factors is None else set(factors)
    x = int(math.sqrt(latent_hw))
    merged = []
    for (factor, layer, head), hm in raw:
        if factor in fset and (head_idx is None or head_idx == head) and (layer_idx is None or layer_idx == layer):
            # GPU autocast(float32) up-casts the bicubic input
            merged.append(F.interpolate(hm.float().unsqueeze(1), size=(x, x), mode='bicubic').clamp_(min=0))
    if not merged:
        if head_idx is not None or layer_idx is not None:
            raise RuntimeError('No heat maps found for the given parameters.')
        raise RuntimeError('No heat maps found. Did you forget to call `with trace(...)` during generation?')
    maps = torch.stack(merged, 0).mean(0)[:, 0]
    if n_rows is not None:
        maps = maps[:n_rows]
    if normalize:
        maps = maps / (maps[1:-1].sum(0, keepdim=True) + 1e-6)
    return maps


# SDXL-base / SD-v1.5 hooked cross-attention layers in locator order:
# (layer_idx, heads, side, head_dim)
def topology(kind: str, latent: Optional[int] = None):
    if kind == 'sdxl':
        base = 128 if latent is None else latent
        a, b = base // 2, base // 4
        return ([(i, 20, b, 64) for i in range(0, 30)] + [(i, 10, a, 64) for i in range(30, 36)] +
                [(i, 10, a, 64) for i in range(36, 40)] + [(i, 20, b, 64) for i in range(40, 60)])
    if kind == 'sd15':
        base = 64 if latent is None else latent
        s = [base // 4, base // 2, base]
        up = [(i, 8, s[i // 3], [160, 80, 40][i // 3]) for i in range(9)]
        down = [(9 + i, 8, [base, base // 2, base // 4][i // 2], [40, 80, 160][i // 2]) for i in range(6)]
        return up + down
    raise ValueError(kind)


def execution_order(layers):
    """UNet forward order: down blocks first, then up blocks (locator lists up first)."""
    n_up = {60: 36, 15: 9}[len(layers)]
    return layers[n_up:] + layers[:n_up]

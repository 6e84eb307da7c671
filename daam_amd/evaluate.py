"""``compute_iou`` / ``compute_ioa`` of the reference's ``daam/evaluate.py:14-35`` on the MI355X: the bicubic resize of the
prediction to the truth mask's size, the binarisation and the three reductions run in one HIP kernel
(``daam_mask_overlap``), for one pair or a whole batch of pairs (the COCO-Gen evaluation loop of ``daam/run/evaluate.py``
calls them once per (word, mask) pair).  Only the final ratio is formed on the host, in fp32 like the reference.

The evaluator classes / mask IO of ``daam/evaluate.py`` (host-side bookkeeping, scipy's Hungarian matching) are not rebuilt.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from . import _native as nat

__all__ = ['compute_iou', 'compute_ioa', 'mask_overlap', 'compute_iou_batch', 'compute_ioa_batch']


def _as_batch(t: torch.Tensor, name: str, device=None) -> torch.Tensor:
    if t.device.type != 'cuda':
        # the reference's evaluation flow hands over CPU masks (load_mask, and expand_as ends in .cpu()): they are moved
        # to the HIP device (the other operand's, else the current one) -- the arithmetic still runs there and only
        # there, and a box without an MI355X fails right here
        if not torch.cuda.is_available():
            raise RuntimeError(f'daam_amd: {name} is a CPU tensor and no HIP device is visible (there is no CPU fallback)')
        t = t.to(device if device is not None else torch.device('cuda', torch.cuda.current_device()))
    if t.dim() == 2:
        t = t.unsqueeze(0)
    if t.dim() != 3:
        raise ValueError(f'{name} must be [h, w] or [n, h, w], got {tuple(t.shape)}')
    return t.to(torch.float32).contiguous()


def mask_overlap(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``a`` [n, ah, aw] (or [ah, aw]) predictions, ``b`` [n, bh, bw] truths -> ``[n, 3]`` fp32 = (sum(a*b), sum(a), sum(b))
    with the reference's preprocessing of ``a``: if ``a.shape[0] != b.shape[0]`` (per pair: the HEIGHTS differ,
    evaluate.py:15) bicubic-resize to ``b``'s size and binarise at 1."""
    dev = a.device if a.device.type == 'cuda' else (b.device if b.device.type == 'cuda' else None)
    a, b = _as_batch(a, 'a', dev), _as_batch(b, 'b', dev)
    if a.shape[0] != b.shape[0]:
        raise ValueError(f'{a.shape[0]} predictions for {b.shape[0]} truth masks')
    if a.device != b.device:
        raise RuntimeError('daam_amd: a and b are on different devices')
    lib = nat.load()
    sums = torch.empty(a.shape[0], 3, dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device):
        nat.check(lib.daam_mask_overlap(a.data_ptr(), a.shape[1], a.shape[2], b.data_ptr(), b.shape[1], b.shape[2], a.shape[0],
                                        sums.data_ptr(), torch.cuda.current_stream(a.device).cuda_stream))
    return sums


def _ratios(sums: torch.Tensor) -> Tuple[np.ndarray, np.ndarray]:
    s = sums.cpu().numpy().astype(np.float32)
    inter, sa, sb = s[:, 0], s[:, 1], s[:, 2]
    eps = np.float32(1e-8)
    union = sa + sb - inter                                   # evaluate.py:21: a.sum() + b.sum() - intersection (fp32)
    return inter / (union + eps), inter / (sa + eps)


def compute_iou_batch(a: torch.Tensor, b: torch.Tensor) -> np.ndarray:
    return _ratios(mask_overlap(a, b))[0]


def compute_ioa_batch(a: torch.Tensor, b: torch.Tensor) -> np.ndarray:
    return _ratios(mask_overlap(a, b))[1]


def compute_iou(a: torch.Tensor, b: torch.Tensor) -> float:
    """evaluate.py:14-23."""
    return float(compute_iou_batch(a, b)[0])


def compute_ioa(a: torch.Tensor, b: torch.Tensor) -> float:
    """evaluate.py:26-35."""
    return float(compute_ioa_batch(a, b)[0])

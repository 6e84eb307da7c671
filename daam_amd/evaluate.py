"""``compute_iou`` / ``compute_ioa`` of the reference's ``daam/evaluate.py:14-35`` on the MI355X: the bicubic resize of the
prediction to the truth mask's size, the binarisation and the three reductions run in one HIP kernel
(``daam_mask_overlap``), for one pair or a whole batch of pairs (the COCO-Gen evaluation loop of ``daam/run/evaluate.py``
calls them once per (word, mask) pair).  Only the final ratio is formed on the host, in fp32 like the reference.

``load_mask`` and the two evaluators of ``daam/evaluate.py:38-117`` (what ``daam/run/evaluate.py`` feeds with these ratios)
are here too: host-side bookkeeping with the reference's names and results; every candidate prediction of one ``log_iou``
call goes to the device in ONE ``daam_mask_overlap`` launch instead of one launch + one host sync per candidate.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Sequence, Tuple, Union

import numpy as np
import torch

from . import _native as nat

__all__ = ['compute_iou', 'compute_ioa', 'mask_overlap', 'compute_iou_batch', 'compute_ioa_batch', 'load_mask',
           'MeanEvaluator', 'UnsupervisedEvaluator']


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


def load_mask(path: str) -> torch.Tensor:
    """A mask stored the way the reference stores them (``*.gt.png`` / ``*.pred.png``, experiment.py:160-163,218-221): an
    RGBA image whose ALPHA channel is the mask; any non-zero alpha counts (evaluate.py:38-43).  Returns a CPU float tensor
    ``[h, w]`` of 0 / 1 -- file IO; ``compute_iou`` / ``compute_ioa`` move it to the device."""
    import PIL.Image
    rgba = np.asarray(PIL.Image.open(path))
    if rgba.ndim != 3 or rgba.shape[2] < 4:
        raise ValueError(f'{path}: not an RGBA mask image (shape {rgba.shape})')
    return torch.from_numpy(np.ascontiguousarray(rgba[:, :, 3] > 0)).to(torch.float32)


def _best_iou(preds: Union[torch.Tensor, Sequence[torch.Tensor]], truth: torch.Tensor) -> float:
    """max over the candidate predictions of IoU(candidate, truth) (evaluate.py:56,88).  Candidates of one shape are
    stacked and scored in one launch; a ragged list is scored shape group by shape group."""
    if isinstance(preds, torch.Tensor):
        preds = [preds]
    preds = list(preds)
    if not preds:
        raise ValueError('max() arg is an empty sequence')                    # what the reference's max() raises
    groups: Dict[tuple, List[torch.Tensor]] = defaultdict(list)
    for p in preds:
        groups[(tuple(p.shape), p.device)].append(p)
    best = -np.inf
    for members in groups.values():
        stack = torch.stack([m.to(torch.float32) for m in members])
        best = max(best, float(compute_iou_batch(stack, truth.unsqueeze(0).expand(len(members), -1, -1)).max()))
    return best


class UnsupervisedEvaluator:
    """evaluate.py:46-82: IoUs logged per (ground-truth segment, predicted segment); ``mean_iou`` matches predicted to
    ground-truth indices with the Hungarian algorithm on the summed-IoU matrix and averages over the matched cells."""

    def __init__(self, name: str = 'UnsupervisedEvaluator'):
        self.name = name
        self.ious = defaultdict(list)
        self.num_samples = 0

    def log_iou(self, preds, truth: torch.Tensor, gt_idx: int = 0, pred_idx: int = 0):
        self.ious[gt_idx].append((pred_idx, _best_iou(preds, truth)))

    @property
    def mean_iou(self) -> float:
        from scipy.optimize import linear_sum_assignment
        entries = [(g, p, v) for g, logged in self.ious.items() for p, v in logged]
        n = max(max(g, p) for g, p, _ in entries) + 1
        total, count = np.zeros((n, n)), np.zeros((n, n))
        for g, p, v in entries:
            total[g, p] += v
            count[g, p] += 1
        rows, cols = linear_sum_assignment(total, maximize=True)
        return total[rows, cols].sum() / count[rows, cols].sum()

    def increment(self):
        self.num_samples += 1

    def __len__(self) -> int:
        return self.num_samples

    def __str__(self):
        return f'{self.name}<{self.mean_iou:.4f} (mIoU) {len(self)} samples>'


class MeanEvaluator:
    """evaluate.py:85-117: running lists of best-candidate IoUs and of mean heat intensities, their means and the
    1.96-sigma half width of the mean IoU."""

    def __init__(self, name: str = 'MeanEvaluator'):
        self.ious: List[float] = []
        self.intensities: List[float] = []
        self.name = name

    def log_iou(self, preds, truth: torch.Tensor) -> 'MeanEvaluator':
        self.ious.append(_best_iou(preds, truth))
        return self

    def log_intensity(self, pred: torch.Tensor) -> 'MeanEvaluator':
        self.intensities.append(pred.mean().item())
        return self

    @property
    def mean_iou(self) -> float:
        return np.mean(self.ious)

    @property
    def mean_intensity(self) -> float:
        return np.mean(self.intensities)

    @property
    def ci95_miou(self) -> float:
        return 1.96 * np.std(self.ious) / np.sqrt(len(self.ious))

    def __len__(self) -> int:
        return max(len(self.ious), len(self.intensities))

    def __str__(self):
        return (f'{self.name}<{self.mean_iou:.4f} (±{self.ci95_miou:.3f} mIoU) {self.mean_intensity:.4f} (mInt) '
                f'{len(self)} samples>')

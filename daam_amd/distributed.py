"""Multi-GPU use of the extraction path: one process per GPU, prompts / seeds sharded across
ranks, no collective on the data path, ONE exchange at the end (``all_gather`` of the final
``[n, tokens, x, x]`` fp32 maps over RCCL -- backend ``nccl`` on ROCm; ``gloo`` on CPU in tests).

The reference has no multi-GPU path: it is single-prompt by construction (``daam/trace.py:172-173``)
and resets its sums at every ``pipe()`` call (``trace.py:179``), which is exactly what makes the
generations independent and the sharding trivial.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

__all__ = ['shard_indices', 'gather_heat_maps', 'trace_prompts']


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin: item i -> rank i mod world_size (keeps ragged tails balanced to within one)."""
    if not 0 <= rank < world_size:
        raise ValueError(f'rank {rank} not in [0, {world_size})')
    return list(range(rank, n_items, world_size))


def gather_heat_maps(local_maps: torch.Tensor, n_items: int, group=None) -> torch.Tensor:
    """``local_maps`` [n_local, ...] holds this rank's items in ``shard_indices`` order.  Returns
    ``[n_items, ...]`` in global item order on every rank.  One ``all_gather_into_tensor`` (ragged
    shards are padded to the largest shard)."""
    if not (dist.is_available() and dist.is_initialized()):
        if local_maps.shape[0] != n_items:
            raise ValueError('single process: local_maps must hold every item')
        return local_maps
    world = dist.get_world_size(group)
    per = -(-n_items // world)
    pad = per - local_maps.shape[0]
    if pad < 0:
        raise ValueError(f'rank holds {local_maps.shape[0]} items, at most {per} expected')
    if pad:
        local_maps = torch.cat([local_maps, local_maps.new_zeros((pad,) + tuple(local_maps.shape[1:]))])
    out = local_maps.new_empty((world * per,) + tuple(local_maps.shape[1:]))
    if local_maps.device.type == 'cuda' and dist.get_backend(group) == 'gloo':
        # gloo moves device tensors through the host anyway; do it explicitly (functional tests with several ranks on
        # one device -- the production backend is nccl = RCCL over xGMI)
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, local_maps.contiguous().cpu(), group=group)
        out.copy_(host)
    else:
        dist.all_gather_into_tensor(out, local_maps.contiguous(), group=group)
    # out[r * per + j] is item r + j * world
    out = out.view((world, per) + tuple(local_maps.shape[1:])).transpose(0, 1).reshape((world * per,) + tuple(local_maps.shape[1:]))
    return out[:n_items]


def trace_prompts(pipe, prompts: Sequence[str], seeds: Optional[Sequence[int]] = None,
                  num_inference_steps: int = 50, trace_kwargs: Optional[dict] = None,
                  compute_kwargs: Optional[dict] = None, group=None,
                  pipe_kwargs: Optional[dict] = None) -> Tuple[torch.Tensor, List[int]]:
    """Run this rank's shard of ``prompts`` through ``pipe`` under ``daam_amd.trace`` and gather the
    global heat maps of ALL prompts (padded to 77 rows) on every rank.  Returns ``(maps
    [n_prompts, 77, x, x], rows)`` with ``rows[i]`` = valid rows (``n_tokens + 2``) of prompt i."""
    from .trace import trace
    from .utils import set_seed
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    mine = shard_indices(len(prompts), rank, world)
    local = []
    with trace(pipe, **(trace_kwargs or {})) as tc:
        for i in mine:
            gen = set_seed(seeds[i]) if seeds is not None else None
            kw = dict(pipe_kwargs or {})
            if gen is not None:
                kw['generator'] = gen
            pipe(prompts[i], num_inference_steps=num_inference_steps, **kw)
            # full 77 rows so that every rank contributes the same shape
            local.append(tc.engine.global_heat_map(**{k: v for k, v in (compute_kwargs or {}).items()
                                                       if k in ('factors', 'head_idx', 'layer_idx')}))
    tokens, side = tc.engine.tokens, tc.engine.out_side
    if local:
        local_maps = torch.stack(local)
    else:
        device = tc.engine.device or torch.device('cuda', torch.cuda.current_device())
        local_maps = torch.zeros(0, tokens, side, side, device=device)
    maps = gather_heat_maps(local_maps, len(prompts), group=group)
    rows = [len(pipe.tokenizer.tokenize(p)) + 2 for p in prompts]
    return maps, rows

"""Hooking framework + cross-attention locator, same behaviour and names as the reference's
``daam/hook.py`` (ObjectHooker :22-73, AggregateHooker :76-86, UNetCrossAttentionLocator
:89-127).  Pure host logic; duck-typed so that neither diffusers nor a GPU is needed to
import it."""
from __future__ import annotations

import functools
from typing import Any, Generic, Iterable, List, Optional, TypeVar

__all__ = ['ObjectHooker', 'ModuleLocator', 'AggregateHooker', 'UNetCrossAttentionLocator']

T = TypeVar('T')


class ModuleLocator(Generic[T]):
    def locate(self, model) -> List[T]:
        raise NotImplementedError


class ObjectHooker(Generic[T]):
    """Context manager that patches attributes of ``module`` on ``hook()`` and restores them
    on ``unhook()``.  Saved originals live in ``old_state`` under ``old_fn_<name>``."""

    _PREFIX = 'old_fn_'

    def __init__(self, module: T):
        self.module: T = module
        self.hooked = False
        self.old_state = {}

    def __enter__(self):
        self.hook()
        return self

    def __exit__(self, exc_type, exc_val, exc_tb):
        self.unhook()

    def hook(self):
        if self.hooked:
            raise RuntimeError('Already hooked module')
        self.old_state = {}
        self.hooked = True
        self._hook_impl()
        return self

    def unhook(self):
        if not self.hooked:
            raise RuntimeError('Module is not hooked')
        for name, original in self.old_state.items():
            if name.startswith(self._PREFIX):
                setattr(self.module, name[len(self._PREFIX):], original)
        self.hooked = False
        self._unhook_impl()
        return self

    def monkey_patch(self, fn_name: str, fn, strict: bool = True):
        """Replace ``module.fn_name`` by ``fn`` bound to the module as first argument.  With
        ``strict=False`` a missing attribute is ignored (SDXL has no ``run_safety_checker``)."""
        try:
            original = getattr(self.module, fn_name)
        except AttributeError:
            if strict:
                raise
            return
        self.old_state[self._PREFIX + fn_name] = original
        setattr(self.module, fn_name, functools.partial(fn, self.module))

    def monkey_super(self, fn_name: str, *args, **kwargs):
        return self.old_state[self._PREFIX + fn_name](*args, **kwargs)

    def _hook_impl(self):
        raise NotImplementedError

    def _unhook_impl(self):
        pass


class AggregateHooker(ObjectHooker[List[ObjectHooker]]):
    def _hook_impl(self):
        for hooker in self.module:
            hooker.hook()

    def _unhook_impl(self):
        for hooker in self.module:
            hooker.unhook()

    def register_hook(self, hook: ObjectHooker):
        self.module.append(hook)


class UNetCrossAttentionLocator(ModuleLocator[Any]):
    """Enumerates ``transformer_block.attn2`` of every block whose class name contains
    ``'CrossAttn'``, visiting ``up_blocks``, then ``down_blocks``, then (optionally) the mid
    block; the position in the returned list is the ``layer_idx`` of the heat-map keys
    (reference hook.py:95-127, trace.py:45,50).  ``restrict`` keeps only these positions
    inside each block; names restart per block (``'{up|down|mid}-attn-{i}'``)."""

    def __init__(self, restrict: Optional[Iterable[int]] = None, locate_middle_block: bool = False):
        self.restrict = restrict
        self.layer_names: List[str] = []
        self.locate_middle_block = locate_middle_block

    def locate(self, model) -> List[Any]:
        self.layer_names.clear()
        stages = [(blk, 'up') for blk in model.up_blocks] + [(blk, 'down') for blk in model.down_blocks]
        if self.locate_middle_block:
            stages.append((model.mid_block, 'mid'))
        found: List[Any] = []
        for block, tag in stages:
            if 'CrossAttn' not in type(block).__name__:
                continue
            attns = [tb.attn2 for st in block.attentions for tb in st.transformer_blocks]
            kept = [a for pos, a in enumerate(attns) if self.restrict is None or pos in self.restrict]
            found.extend(kept)
            self.layer_names.extend(f'{tag}-attn-{i}' for i in range(len(kept))
                                    if self.restrict is None or i in self.restrict)
        return found

"""Attribute patching with guaranteed restore, and the cross-attention locator.

Public names and behaviour follow the reference's ``daam/hook.py`` (``ObjectHooker`` :22-73,
``AggregateHooker`` :76-86, ``UNetCrossAttentionLocator`` :89-127) because ``daam_amd`` is a drop-in:
``hook()`` / ``unhook()`` raise ``RuntimeError`` on misuse, ``monkey_patch`` binds the replacement to the
patched object, the locator defines ``layer_idx``.  The implementation is a small journal of
``(attribute, original value)`` records that is replayed backwards on ``unhook()``.
Pure host logic, duck-typed: neither diffusers nor a GPU is needed to import it.
"""
from __future__ import annotations

import functools
from typing import Any, Callable, Dict, Generic, Iterable, Iterator, List, Optional, Tuple, TypeVar

__all__ = ['ObjectHooker', 'ModuleLocator', 'AggregateHooker', 'UNetCrossAttentionLocator']

T = TypeVar('T')

_KEY = 'old_fn_{}'          # key format of ``old_state`` (kept for code that pokes at it like the reference's)


class ModuleLocator(Generic[T]):
    """Finds the objects of a model that a hooker should be attached to."""

    def locate(self, model) -> List[T]:
        raise NotImplementedError


class ObjectHooker(Generic[T]):
    """Reversible patching of one object (``self.module``).

    Subclasses patch in ``_hook_impl`` (usually through ``monkey_patch``) and may undo extra state in
    ``_unhook_impl``.  Usable as a context manager."""

    def __init__(self, module: T):
        self.module: T = module
        self.hooked: bool = False
        self._journal: List[Tuple[str, Any]] = []      # (attribute name, value before the patch), in patch order

    # -- reference-compatible view of the journal ------------------------------------------------
    @property
    def old_state(self) -> Dict[str, Any]:
        return {_KEY.format(name): value for name, value in self._journal}

    # -- lifecycle -------------------------------------------------------------------------------
    def hook(self):
        if self.hooked:
            raise RuntimeError('Already hooked module')
        self._journal = []
        self.hooked = True
        self._hook_impl()
        return self

    def unhook(self):
        if not self.hooked:
            raise RuntimeError('Module is not hooked')
        while self._journal:
            name, value = self._journal.pop()
            setattr(self.module, name, value)
        self.hooked = False
        self._unhook_impl()
        return self

    def __enter__(self):
        return self.hook()

    def __exit__(self, *exc_info):
        self.unhook()

    # -- patching --------------------------------------------------------------------------------
    def monkey_patch(self, fn_name: str, fn: Callable, strict: bool = True) -> None:
        """``module.fn_name = partial(fn, module)``, remembering what was there.  A missing attribute is an
        ``AttributeError`` unless ``strict=False`` (SDXL pipelines have no ``run_safety_checker``)."""
        if not hasattr(self.module, fn_name):
            if strict:
                raise AttributeError(f'{type(self.module).__name__!r} object has no attribute {fn_name!r}')
            return
        self._journal.append((fn_name, getattr(self.module, fn_name)))
        setattr(self.module, fn_name, functools.partial(fn, self.module))

    def monkey_super(self, fn_name: str, *args, **kwargs):
        """Call what ``fn_name`` was before this hooker patched it."""
        for name, value in reversed(self._journal):
            if name == fn_name:
                return value(*args, **kwargs)
        raise KeyError(_KEY.format(fn_name))

    def _hook_impl(self) -> None:
        raise NotImplementedError

    def _unhook_impl(self) -> None:
        pass


class AggregateHooker(ObjectHooker[List[ObjectHooker]]):
    """A hooker over a list of hookers: hooks / unhooks them in list order."""

    def register_hook(self, hook: ObjectHooker) -> None:
        self.module.append(hook)

    def _hook_impl(self) -> None:
        for member in self.module:
            member.hook()

    def _unhook_impl(self) -> None:
        for member in self.module:
            member.unhook()


class UNetCrossAttentionLocator(ModuleLocator[Any]):
    """Enumerates ``transformer_block.attn2`` of every block whose class name contains ``'CrossAttn'``:
    ``up_blocks`` first, then ``down_blocks``, then (optionally) the mid block.  The position in the
    returned list is the ``layer_idx`` of the heat-map keys (reference hook.py:95-127, trace.py:45,50).
    ``restrict`` keeps only those positions inside each block; names restart per block
    (``'{up|down|mid}-attn-{i}'``, so they are not unique across blocks -- reference hook.py:123)."""

    def __init__(self, restrict: Optional[Iterable[int]] = None, locate_middle_block: bool = False):
        self.restrict = restrict
        self.locate_middle_block = locate_middle_block
        self.layer_names: List[str] = []

    def _stages(self, unet) -> Iterator[Tuple[str, Any]]:
        yield from (('up', block) for block in unet.up_blocks)
        yield from (('down', block) for block in unet.down_blocks)
        if self.locate_middle_block:
            yield 'mid', unet.mid_block

    def _wanted(self, position: int) -> bool:
        return self.restrict is None or position in self.restrict

    def locate(self, model) -> List[Any]:
        self.layer_names.clear()
        located: List[Any] = []
        for tag, block in self._stages(model):
            if 'CrossAttn' not in type(block).__name__:
                continue
            cross = [tb.attn2 for transformer in block.attentions for tb in transformer.transformer_blocks]
            kept = [attn for position, attn in enumerate(cross) if self._wanted(position)]
            located += kept
            # the reference numbers the names over the KEPT list but filters them with `restrict` again
            self.layer_names += [f'{tag}-attn-{i}' for i in range(len(kept)) if self._wanted(i)]
        return located

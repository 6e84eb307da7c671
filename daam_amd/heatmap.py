"""Heat-map containers with the reference's names (``daam/heatmap.py``): the raw
per-(factor, layer, head) collection -- here a zero-copy view over the device running sums --
and ``GlobalHeatMap`` / ``WordHeatMap``, whose arithmetic runs in ``libdaam_hip.so``."""
from __future__ import annotations

from dataclasses import dataclass
from functools import lru_cache
from typing import Any, Iterable, Iterator, Optional, Set, Tuple

import torch

from . import engine as _engine
from .utils import cached_nlp, compute_token_merge_indices

__all__ = ['GlobalHeatMap', 'RawHeatMapCollection', 'WordHeatMap', 'ParsedHeatMap', 'SyntacticHeatMapPair']

RawHeatMapKey = Tuple[int, int, int]  # factor, layer, head


class RawHeatMapCollection:
    """Same surface as reference heatmap.py:148-172 (``update / factors / layers / heads /
    __iter__ / clear``).  Iteration yields ``((factor, layer, head), Tensor[tokens, h, w])`` in
    first-update order; the tensors are views of the live device buffers."""

    def __init__(self, engine: '_engine.HeatMapEngine'):
        self._engine = engine

    def update(self, factor: int, layer_idx: int, head_idx: int, heatmap: torch.Tensor):
        self._engine.add_map(factor, layer_idx, head_idx, heatmap)

    def factors(self) -> Set[int]:
        return {k[0] for k in self._engine.keys()}

    def layers(self) -> Set[int]:
        return {k[1] for k in self._engine.keys()}

    def heads(self) -> Set[int]:
        return {k[2] for k in self._engine.keys()}

    def __iter__(self) -> Iterator[Tuple[RawHeatMapKey, torch.Tensor]]:
        return self._engine.items()

    def __len__(self) -> int:
        return len(self._engine.keys())

    def clear(self):
        self._engine.clear()


class WordHeatMap:
    """reference heatmap.py:56-96."""

    def __init__(self, heatmap: torch.Tensor, word: Optional[str] = None, word_idx: Optional[int] = None):
        self.word = word
        self.word_idx = word_idx
        self.heatmap = heatmap

    @property
    def value(self) -> torch.Tensor:
        return self.heatmap

    def expand_as(self, image, absolute: bool = False, threshold: Optional[float] = None, plot: bool = False,
                  **plot_kwargs) -> torch.Tensor:
        """Bicubic to ``(image.size[0], image.size[1])`` (PIL order, as the reference passes
        it, heatmap.py:80), min-max normalise unless ``absolute``, optional threshold; returns a
        CPU tensor like the reference (heatmap.py:88)."""
        out = _engine.expand_word_map(self.heatmap.float(), int(image.size[0]), int(image.size[1]),
                                      absolute=absolute, threshold=threshold)
        out = out.cpu()
        if plot:
            self.plot_overlay(image, **plot_kwargs)
        return out

    def compute_ioa(self, other: 'WordHeatMap') -> float:
        """reference heatmap.py:95-96."""
        from .evaluate import compute_ioa
        return compute_ioa(self.heatmap, other.heatmap)

    def plot_overlay(self, image, out_file=None, color_normalize=True, ax=None, **expand_kwargs):
        from .plotting import plot_overlay_heat_map
        plot_overlay_heat_map(image, self.expand_as(image, **expand_kwargs), word=self.word, out_file=out_file,
                              color_normalize=color_normalize, ax=ax)


@dataclass
class SyntacticHeatMapPair:
    """A dependency arc of the prompt with the word maps at both ends (heatmap.py:99-105)."""
    head_heat_map: WordHeatMap
    dep_heat_map: WordHeatMap
    head_text: str
    dep_text: str
    relation: str


@dataclass
class ParsedHeatMap:
    """A parsed prompt token with its word map (heatmap.py:108-111); ``token`` is the parser's token object."""
    word_heat_map: WordHeatMap
    token: Any


class GlobalHeatMap:
    """reference heatmap.py:114-142."""

    def __init__(self, tokenizer: Any, prompt: str, heat_maps: torch.Tensor):
        self.tokenizer = tokenizer
        self.heat_maps = heat_maps
        self.prompt = prompt
        self.compute_word_heat_map = lru_cache(maxsize=50)(self.compute_word_heat_map)

    def compute_word_heat_map(self, word: str, word_idx: Optional[int] = None, offset_idx: int = 0) -> WordHeatMap:
        merge_idxs, word_idx = compute_token_merge_indices(self.tokenizer, self.prompt, word, word_idx, offset_idx)
        return WordHeatMap(_engine.word_heat_map(self.heat_maps, merge_idxs), word, word_idx)

    def parsed_heat_maps(self) -> Iterable[ParsedHeatMap]:
        """One ``ParsedHeatMap`` per token of the parsed prompt whose text is found among the prompt's tokenizer tokens
        (heatmap.py:125-131; the others are skipped)."""
        for token in cached_nlp(self.prompt):
            try:
                yield ParsedHeatMap(self.compute_word_heat_map(token.text), token)
            except ValueError:
                continue

    def dependency_relations(self) -> Iterable[SyntacticHeatMapPair]:
        """One pair per non-root token: the maps of the token and of its syntactic head (heatmap.py:133-142)."""
        for token in cached_nlp(self.prompt):
            if token.dep_ == 'ROOT':
                continue
            try:
                dependent = self.compute_word_heat_map(token.text)
                head = self.compute_word_heat_map(token.head.text)
            except ValueError:
                continue
            yield SyntacticHeatMapPair(head, dependent, token.head.text, token.text, token.dep_)

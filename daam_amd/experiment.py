"""``GenerationExperiment``: the on-disk result of one traced generation, in the layout the
reference writes (``daam/experiment.py:102-167,303-344``):

    <path>/<id>/prompt.txt, seed.txt, annotations.json
    <path>/<id>/<subtype>/generation.pt   (the pickled dataclass, incl. the [n_tok+2, x, x] map)
    <path>/<id>/<subtype>/output.png, <word>.heat_map.png

Downstream of the extraction path: host-side persistence only.  The
COCO label tables, ground-truth / prediction mask IO and the evaluation helpers of the reference
are not rebuilt."""
from __future__ import annotations

import json
import warnings
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, Optional, Union

import numpy as np
import torch

__all__ = ['GenerationExperiment']


@dataclass
class GenerationExperiment:
    """Holds the parameters and results of one generation.  Pickleable."""
    image: Any
    global_heat_map: torch.Tensor
    prompt: str

    seed: Optional[int] = None
    id: str = '.'
    path: Optional[Path] = None

    truth_masks: Optional[Dict[str, torch.Tensor]] = None
    prediction_masks: Optional[Dict[str, torch.Tensor]] = None
    annotations: Optional[Dict[str, Any]] = None
    subtype: Optional[str] = '.'
    tokenizer: Any = None

    def __post_init__(self):
        if isinstance(self.path, str):
            self.path = Path(self.path)
        self.path = None if self.path is None else self.path / self.id

    def nsfw(self) -> bool:
        return np.sum(np.array(self.image)) == 0

    def heat_map(self, tokenizer=None):
        """The experiment's ``GlobalHeatMap`` (reference experiment.py:240).  A checkpoint holds the map on the CPU;
        word maps are computed by ``libdaam_hip.so``, so the map goes (back) to the HIP device here."""
        from .heatmap import GlobalHeatMap
        maps = self.global_heat_map
        if maps.device.type != 'cuda' and torch.cuda.is_available():
            maps = maps.to('cuda', torch.float32).contiguous()     # without a device the word-map call fails loudly
        return GlobalHeatMap(self.tokenizer if tokenizer is None else tokenizer, self.prompt, maps)

    def clear_checkpoint(self):
        (self.path / self.subtype / 'generation.pt').unlink(missing_ok=True)

    def save(self, path: Union[str, Path, None] = None, heat_maps: bool = True, tokenizer=None):
        root = self.path if path is None else Path(path) / self.id
        tokenizer = self.tokenizer if tokenizer is None else tokenizer
        sub = root / self.subtype
        sub.mkdir(parents=True, exist_ok=True)
        # the map is stored on the CPU so that the checkpoint loads anywhere
        on_disk = GenerationExperiment.__new__(GenerationExperiment)
        on_disk.__dict__.update(self.__dict__)
        on_disk.global_heat_map = self.global_heat_map.detach().cpu()
        torch.save(on_disk, sub / 'generation.pt')
        if hasattr(self.image, 'save'):
            self.image.save(sub / 'output.png')
        (root / 'prompt.txt').write_text(self.prompt)
        (root / 'seed.txt').write_text(str(self.seed))
        if heat_maps and tokenizer is not None and (self.global_heat_map.device.type == 'cuda' or torch.cuda.is_available()):
            self.save_all_heat_maps(tokenizer)
        self.save_annotations(root)

    def save_annotations(self, path: Optional[Path] = None):
        path = self.path if path is None else path
        if self.annotations is not None:
            with (path / 'annotations.json').open('w') as f:
                json.dump(self.annotations, f)

    def annotate(self, key: str, value: Any) -> 'GenerationExperiment':
        if self.annotations is None:
            self.annotations = {}
        self.annotations[key] = value
        return self

    def save_heat_map(self, word: str, tokenizer=None, crop: Optional[int] = None, output_prefix: str = '',
                      absolute: bool = False) -> Path:
        path = self.path / self.subtype / f'{output_prefix}{word.lower()}.heat_map.png'
        self.heat_map(tokenizer).compute_word_heat_map(word).plot_overlay(
            self.image, out_file=path, color_normalize=not absolute, crop=crop)
        return path

    def save_all_heat_maps(self, tokenizer=None, crop: Optional[int] = None) -> Dict[str, Path]:
        out = {}
        for word in self.prompt.split(' '):
            try:
                out[word] = self.save_heat_map(word, tokenizer, crop=crop)
            except Exception as exc:   # noqa: BLE001 -- the reference skips a word on ANY failure (bare except, experiment.py:250-255)
                # "Search word ... not found in prompt!" is the expected one; anything else is skipped too (one bad word must
                # not abort save()), but said out loud
                if not isinstance(exc, ValueError):
                    warnings.warn(f'daam_amd: heat map of {word!r} skipped: {type(exc).__name__}: {exc}')
        return out

    @staticmethod
    def read_seed(path: Union[str, Path], prompt_id: Optional[str] = None) -> int:
        path = Path(path) if prompt_id is None else Path(path) / prompt_id
        return int((path / 'seed.txt').read_text().strip())

    @staticmethod
    def read_prompt(path: Union[str, Path], prompt_id: Optional[str] = None) -> str:
        path = Path(path) if prompt_id is None else Path(path) / prompt_id
        return (path / 'prompt.txt').read_text().strip()

    @staticmethod
    def has_experiment(path: Union[str, Path], prompt_id: str) -> bool:
        return (Path(path) / prompt_id / 'generation.pt').exists()

    @classmethod
    def load(cls, path: Union[str, Path], subtype: str = '.') -> 'GenerationExperiment':
        path = Path(path)
        exp = torch.load(path / subtype / 'generation.pt', weights_only=False)
        exp.subtype = subtype
        exp.path = path
        ann = path / 'annotations.json'
        exp.annotations = json.loads(ann.read_text()) if ann.exists() else None
        return exp

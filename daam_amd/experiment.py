"""``GenerationExperiment``: the on-disk result of one traced generation, in the layout the
reference writes (``daam/experiment.py:102-167,303-344``):

    <path>/<id>/prompt.txt, seed.txt, annotations.json, <word>.gt.png (ground-truth masks)
    <path>/<id>/<subtype>/generation.pt   (the pickled dataclass, incl. the [n_tok+2, x, x] map)
    <path>/<id>/<subtype>/output.png, <word>.heat_map.png
    <path>/<id>/<subtype>/<word>.<name>.pred.png, composite.<name>.pred.png (predicted masks)

Downstream of the extraction path: host-side persistence only.  Masks are RGBA PNGs with the
mask in every channel (``evaluate.load_mask`` reads the alpha channel); a composite prediction is one index image decoded
with a vocabulary.  Checkpoints written by the reference itself (pickled as ``daam.experiment.GenerationExperiment``) load
here: the unpickler maps the ``daam`` package onto this one."""
from __future__ import annotations

import json
import pickle
import warnings
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from .coco import (COCO80_INDICES, COCO80_LABELS, COCO80_ONTOLOGY, COCO80_TO_27, COCOSTUFF27_LABELS,  # noqa: F401
                   UNUSED_LABELS, build_word_list_coco80)
from .evaluate import load_mask

__all__ = ['GenerationExperiment', 'COCO80_LABELS', 'COCOSTUFF27_LABELS', 'COCO80_INDICES', 'build_word_list_coco80']


class _ReferencePickle:
    """``pickle_module`` for ``torch.load``: a checkpoint the reference wrote names its classes ``daam.<module>.<Class>``;
    they resolve to the classes of this package (same fields -- ``GenerationExperiment`` is the drop-in)."""
    __name__ = 'daam_amd.experiment._ReferencePickle'

    class Unpickler(pickle.Unpickler):
        def find_class(self, module: str, name: str):
            if module == 'daam' or module.startswith('daam.'):
                module = 'daam_amd' + module[len('daam'):]
            return super().find_class(module, name)

    load = staticmethod(pickle.load)
    loads = staticmethod(pickle.loads)
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)
    Pickler = pickle.Pickler
    PicklingError = pickle.PicklingError
    UnpicklingError = pickle.UnpicklingError
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    DEFAULT_PROTOCOL = pickle.DEFAULT_PROTOCOL


def _mask_image(mask: torch.Tensor):
    """A [h, w] mask of 0 .. 1 as the RGBA image the reference writes: the byte ``mask * 255`` in all four channels
    (experiment.py:161,219)."""
    import PIL.Image
    plane = (mask.detach().to('cpu') * 255).to(torch.uint8).numpy()
    return PIL.Image.fromarray(np.repeat(plane[:, :, None], 4, axis=2))


def _add_mask(masks: Dict[str, torch.Tensor], word: str, mask: torch.Tensor, simplify80: bool = False) -> Dict[str, torch.Tensor]:
    """File ``mask`` under ``word`` (under its coarse COCO-Stuff name with ``simplify80``); masks that land on one name are
    united (sum clamped to [0, 1]) -- experiment.py:94-104.  Names arrive lower-cased from the callers."""
    if simplify80:
        word = COCO80_TO_27.get(word, word)
    if word in masks:
        masks[word] = (masks[word.lower()] + mask).clamp_(0, 1)
    else:
        masks[word] = mask
    return masks


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
        """Reference layout on disk (experiment.py:303-344).  One-directional compatibility: ``load`` reads checkpoints pickled by
        the reference (``daam.experiment.GenerationExperiment`` resolves to this class), but ``generation.pt`` written HERE names
        ``daam_amd.experiment.GenerationExperiment`` -- the unmodified reference cannot unpickle it without this package importable
        (every other file of the directory -- prompt.txt, seed.txt, the PNGs, annotations.json -- is the reference's own format)."""
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
        for name, mask in (self.truth_masks or {}).items():
            _mask_image(mask).save(root / f'{name.lower()}.gt.png')
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

    # ---- ground-truth / predicted masks (experiment.py:170-221) ----
    def _load_truth_masks(self, simplify80: bool = False) -> Dict[str, torch.Tensor]:
        masks: Dict[str, torch.Tensor] = {}
        for file in self.path.glob('*.gt.png'):
            _add_mask(masks, file.name.split('.gt.png')[0].lower(), load_mask(str(file)), simplify80)
        return masks

    def _load_pred_masks(self, pred_prefix: str, composite: bool = False, simplify80: bool = False,
                         vocab: Optional[Sequence[str]] = None) -> Dict[str, torch.Tensor]:
        """``<word>.<pred_prefix>.pred.png`` files of the subtype directory, or -- ``composite`` -- the one index image
        ``composite.<pred_prefix>.pred.png`` split into a mask per pixel value, named by ``vocab[value]``."""
        import PIL.Image
        masks: Dict[str, torch.Tensor] = {}
        sub = self.path / self.subtype
        if composite:
            names = UNUSED_LABELS if vocab is None else vocab
            file = sub / f'composite.{pred_prefix}.pred.png'
            if file.exists():
                index = np.asarray(PIL.Image.open(file))
                for value in np.unique(index):
                    _add_mask(masks, names[value], torch.from_numpy((index == value).astype(np.float32)), simplify80)
        else:
            marker = f'.{pred_prefix}.pred'
            for file in sub.glob(f'*{marker}.png'):
                _add_mask(masks, file.name.split(marker)[0].lower(), load_mask(str(file)), simplify80)
        return masks

    def clear_prediction_masks(self, name: str):
        for file in (self.path / self.subtype).glob(f'*.{name}.pred.png'):
            file.unlink()

    def save_prediction_mask(self, mask: torch.Tensor, word: str, name: str):
        _mask_image(mask).save(self.path / self.subtype / f'{word.lower()}.{name}.pred.png')

    @staticmethod
    def contains_truth_mask(path: Union[str, Path], prompt_id: Optional[str] = None) -> bool:
        path = Path(path) if prompt_id is None else Path(path) / prompt_id
        return any(path.glob('*.gt.png'))

    @staticmethod
    def has_annotations(path: Union[str, Path]) -> bool:
        return (Path(path) / 'annotations.json').exists()

    def _try_load_annotations(self) -> Optional[Dict[str, Any]]:
        file = self.path / 'annotations.json'
        return json.loads(file.read_text()) if file.exists() else None

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
    def load(cls, path: Union[str, Path], pred_prefix: str = 'daam', composite: bool = False, simplify80: bool = False,
             vocab: Optional[Sequence[str]] = None, subtype: str = '.', all_subtypes: bool = False
             ) -> Union['GenerationExperiment', List['GenerationExperiment']]:
        """experiment.py:303-344: the checkpoint of ``subtype`` with its masks and annotations read back from the files
        around it; ``all_subtypes`` = one experiment per sub-directory that holds a readable checkpoint."""
        path = Path(path)
        if all_subtypes:
            found = []
            for directory in path.iterdir():
                if directory.is_dir():
                    try:
                        found.append(cls.load(path, pred_prefix=pred_prefix, composite=composite, simplify80=simplify80,
                                              vocab=vocab, subtype=directory.name))
                    except Exception:   # noqa: BLE001 -- not an experiment directory: skipped, like the reference (:330-331)
                        pass
            return found
        exp = torch.load(path / subtype / 'generation.pt', weights_only=False, pickle_module=_ReferencePickle)
        exp.subtype = subtype
        exp.path = path
        exp.truth_masks = exp._load_truth_masks(simplify80=simplify80)
        exp.prediction_masks = exp._load_pred_masks(pred_prefix, composite=composite, simplify80=simplify80, vocab=vocab)
        exp.annotations = exp._try_load_annotations()
        return exp

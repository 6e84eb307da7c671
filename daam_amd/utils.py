"""Host helpers next to the extraction path, with the names and behaviour of the reference's
``daam/utils.py`` (device / autocast policy :22-36, seeding :46-55, cache directory :58-70, prompt-word to
heat-map-row mapping :73-91, the cached spaCy parse :94-109, the thresholded-mask plot :39-43).  spaCy and matplotlib are
imported only by the two helpers that need them."""
from __future__ import annotations

import os
import random
import sys
from pathlib import Path
from functools import lru_cache
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

__all__ = ['set_seed', 'compute_token_merge_indices', 'plot_mask_heat_map', 'cached_nlp', 'set_nlp', 'cache_dir',
           'auto_device', 'auto_autocast']

_EOW = '</w>'        # CLIP BPE end-of-word marker


def _gpu() -> bool:
    return torch.cuda.is_available()


def auto_device(obj=torch.device('cpu')):
    """A ``torch.device`` argument answers "which device should I use"; anything else (module, tensor) is
    moved to the GPU when there is one."""
    if isinstance(obj, torch.device):
        return torch.device('cuda') if _gpu() else torch.device('cpu')
    return obj.to('cuda') if _gpu() else obj


def auto_autocast(*args, **kwargs):
    """``torch.autocast('cuda', ...)`` that is a no-op on a machine without a GPU.  On the GPU it is what puts
    the reference's bicubic resize on the fp32 policy; the kernels here hard-wire that."""
    if not _gpu():
        kwargs = dict(kwargs, enabled=False)
    return torch.autocast('cuda', *args, **kwargs)


def set_seed(seed: int) -> torch.Generator:
    """Seed python / numpy / torch (all devices) and hand back a seeded generator on the default device --
    the object the reference passes as ``generator=`` to the pipeline."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if _gpu():
        torch.cuda.manual_seed_all(seed)
    return torch.Generator(device=auto_device()).manual_seed(seed)


def cache_dir() -> Path:
    """Per-user cache directory ``.../daam`` (XDG on Linux / BSD, ``~/Library/Caches`` on macOS,
    ``%LOCALAPPDATA%`` on Windows) -- where ``save_heads`` / ``load_heads`` keep their files by default."""
    home = Path(os.path.expanduser('~'))
    if sys.platform == 'darwin':
        base = home / 'Library' / 'Caches'
    elif os.name == 'posix':
        base = Path(os.environ.get('XDG_CACHE_HOME') or home / '.cache')
    else:
        base = Path(os.environ.get('LOCALAPPDATA') or home / 'AppData' / 'Local')
    return base / 'daam'


def _bare_tokens(tokenizer, text: str) -> List[str]:
    return [piece.replace(_EOW, '') for piece in tokenizer.tokenize(text)]


def _occurrences(haystack: Sequence[str], needle: Sequence[str]) -> List[int]:
    n = len(needle)
    return [i for i in range(len(haystack)) if list(haystack[i:i + n]) == list(needle)]


def compute_token_merge_indices(tokenizer, prompt: str, word: str, word_idx: Optional[int] = None,
                                offset_idx: int = 0) -> Tuple[List[int], Optional[int]]:
    """Rows of a global heat map that belong to ``word``.

    Row 0 is the start-of-text token, so prompt token ``i`` lives in row ``i + 1``.  The word is matched as a
    token sequence (case-folded, end-of-word markers stripped) at every position of the prompt; every
    occurrence contributes all of its sub-word rows, shifted by ``offset_idx``.  ``word_idx`` names the prompt
    token directly and skips the search.  A word that does not occur is a ``ValueError`` (same message as the
    reference)."""
    if word_idx is not None:
        return [word_idx + 1], word_idx
    word = word.lower()
    needle = _bare_tokens(tokenizer, word)
    starts = _occurrences(_bare_tokens(tokenizer, prompt.lower()), needle)
    rows = [start + offset_idx + j + 1 for start in starts for j in range(len(needle))]
    if not rows:
        raise ValueError(f'Search word {word} not found in prompt!')
    return rows, word_idx


def plot_mask_heat_map(im, heat_map: torch.Tensor, threshold: float = 0.4):
    """Show the image with everything outside ``heat_map > threshold`` blacked out (utils.py:39-43)."""
    from matplotlib import pyplot as plt
    keep = (heat_map.detach().squeeze().to('cpu', torch.float32) > threshold).to(torch.float32)
    pixels = torch.from_numpy(np.array(im)).to(torch.float32) / 255
    plt.imshow(pixels * keep.unsqueeze(-1))


_parsers: dict = {}        # spaCy pipeline name -> loaded pipeline (or whatever ``set_nlp`` registered)


def set_nlp(parser: Optional[Callable], type: str = 'en_core_web_md') -> None:
    """Register the callable ``prompt -> parsed document`` that ``cached_nlp`` uses for pipeline ``type`` (``None`` forgets
    it): a spaCy pipeline loaded elsewhere, or any parser whose tokens carry ``.text`` / ``.dep_`` / ``.head``."""
    if parser is None:
        _parsers.pop(type, None)
    else:
        _parsers[type] = parser
    cached_nlp.cache_clear()


@lru_cache(maxsize=100000)
def cached_nlp(prompt: str, type: str = 'en_core_web_md'):
    """The spaCy parse of ``prompt``, cached per (prompt, pipeline) like the reference's (utils.py:97-109).  The pipeline is
    loaded on first use; a missing model is an error that names it (the reference shells out to ``spacy download`` --
    nothing is downloaded here)."""
    parser = _parsers.get(type)
    if parser is None:
        try:
            import spacy
        except ImportError as exc:
            raise ImportError('daam_amd: parsed_heat_maps / dependency_relations need spaCy (or a parser registered with '
                              'daam_amd.utils.set_nlp)') from exc
        try:
            parser = spacy.load(type)
        except OSError as exc:
            raise OSError(f'daam_amd: spaCy pipeline {type!r} is not installed (python -m spacy download {type})') from exc
        _parsers[type] = parser
    return parser(prompt)

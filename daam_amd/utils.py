"""Host utilities on / next to the hot path (reference ``daam/utils.py``): autocast policy,
seeding, device choice, cache dir, token merge indices.  No spaCy / matplotlib here."""
from __future__ import annotations

import os
import random
import sys
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np
import torch

__all__ = ['set_seed', 'compute_token_merge_indices', 'cache_dir', 'auto_device', 'auto_autocast']


def auto_device(obj=torch.device('cpu')):
    """reference utils.py:22-29."""
    if isinstance(obj, torch.device):
        return torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    return obj.to('cuda') if torch.cuda.is_available() else obj


def auto_autocast(*args, **kwargs):
    """reference utils.py:32-36: autocast, force-disabled without a GPU."""
    if not torch.cuda.is_available():
        kwargs['enabled'] = False
    return torch.autocast('cuda', *args, **kwargs)


def set_seed(seed: int) -> torch.Generator:
    """reference utils.py:46-55."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    gen = torch.Generator(device=auto_device())
    gen.manual_seed(seed)
    return gen


def cache_dir() -> Path:
    """reference utils.py:58-70."""
    if os.name == 'posix' and sys.platform != 'darwin':
        return Path(os.environ.get('XDG_CACHE_HOME', os.path.expanduser('~/.cache')), 'daam')
    if sys.platform == 'darwin':
        return Path(os.path.expanduser('~'), 'Library/Caches/daam')
    local = os.environ.get('LOCALAPPDATA') or os.path.expanduser('~\\AppData\\Local')
    return Path(local, 'daam')


def compute_token_merge_indices(tokenizer, prompt: str, word: str, word_idx: Optional[int] = None,
                                offset_idx: int = 0) -> Tuple[List[int], Optional[int]]:
    """Rows of the global heat map that belong to ``word`` (reference utils.py:73-91):
    lower-case, strip the ``</w>`` end-of-word marker, match the word's token sequence at
    every position of the prompt, and shift by one for the SOS row.  ``word_idx`` bypasses the
    search.  Raises ``ValueError`` when the word does not occur."""
    if word_idx is not None:
        return [word_idx + 1], word_idx

    def pieces(text: str) -> List[str]:
        return [tok.replace('</w>', '') for tok in tokenizer.tokenize(text)]

    prompt_toks = pieces(prompt.lower())
    word = word.lower()
    word_toks = pieces(word)
    n = len(word_toks)
    rows: List[int] = []
    for start in range(len(prompt_toks)):
        if prompt_toks[start:start + n] == word_toks:
            rows.extend(start + offset_idx + j for j in range(n))
    if not rows:
        raise ValueError(f'Search word {word} not found in prompt!')
    return [r + 1 for r in rows], word_idx

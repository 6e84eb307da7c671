import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, f'{name}.npz'), allow_pickle=False)
    meta = json.loads(str(z['meta']))
    return z, meta


def golden_pipe(meta, device='cpu', seed_offset=0):
    """Rebuild the fake pipeline a golden case was generated with (same seeds -> same bits).  ``seed_offset=100`` is
    the second pipeline of the save_heads / load_heads cases (other hidden states, other V)."""
    import torch
    from oracle import fake_diffusers as fd
    dtype = getattr(torch, meta['dtype'])
    return fd.make_pipe(meta['kind'], device=device, dtype=dtype, batch=meta['batch'], seed=meta['seed'] + seed_offset,
                        mini=meta.get('mini', True), identity_proj=True, **meta['unet'])


GOLDEN_CASES = ['sd15_f32', 'sd15_f16', 'sd15_bf16', 'sdxl_f32', 'sdxl_f16', 'sdxl2048_f32', 'sd15_nocfg_f32', 'sd15_b4_f32',
                'sdxl_heads_f32', 'sd15_heads_f16', 'sd15_upcast_attn_f16', 'sd15_upcast_softmax_f16', 'sd15_real_f16',
                'sdxl_real_f16']


@pytest.fixture(params=GOLDEN_CASES)
def golden_case(request):
    z, meta = load_golden(request.param)
    return request.param, z, meta

"""GPU parity: the HIP path (through the C ABI, libdaam_hip.so) against the numpy oracle and the
golden vectors produced by executing the unmodified reference.  Run with ``-m gpu`` on an
MI355X.  Tolerances are stated next to each comparison:

  * fp32 pipeline, fp32 sums ........ <= 2e-6 * max|ref|  (summation order only)
  * fp16 pipeline, fp16 sums ('exact') vs the literal fp16 reference:
        raw running sums ............ <= 1 ulp of the sum (a logit straddling an fp16 rounding
                                      boundary moves one probability by one ulp)
        global heat maps ............ <= 1e-3 max-abs (BASELINE.json north_star); observed ~1e-4
  * bf16 pipeline, bf16 sums: the same statements with 8-bit significands (1 ulp = 2^-8 relative); global
        heat maps <= 1e-2 max-abs (the bf16 reference differs from the fp32 one by that order itself)
  * probabilities path (tap='probs') . bit-exact running sums (same adds, same order)
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, ROOT, golden_pipe, load_golden
from oracle import heatmap_oracle as ho
from oracle.make_golden import OUT_SAMPLE_ROWS, SAMPLE_TOKENS

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _engine(n_layers=4, **kw):
    from daam_amd.engine import HeatMapEngine
    return HeatMapEngine(n_layers, tokens=77, out_side=kw.pop('out_side', 64), **kw)


def _qk(rng, batch, heads, hw, d, dtype, sos_gain=3.0):
    """dtype: a numpy dtype, or ho.BF16 (then float32 arrays holding bf16-representable values)."""
    q = rng.standard_normal((batch, hw, heads * d)).astype(np.float32)
    k = rng.standard_normal((batch, 77, heads * d)).astype(np.float32)
    k[:, 0, :] *= sos_gain
    if ho.is_bf16(dtype):
        return ho.round_bf16(q), ho.round_bf16(k)
    return q.astype(dtype), k.astype(dtype)


def _dev(x, dtype=None):
    """numpy -> device tensor; bf16 data travels as float32 and is narrowed (exactly) on the device."""
    t = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    return t.to(torch.bfloat16) if ho.is_bf16(dtype) else t


def _to_bh(x, heads):
    b, s, c = x.shape
    d = c // heads
    return np.ascontiguousarray(x.reshape(b, s, heads, d).transpose(0, 2, 1, 3)).reshape(b * heads, s, d)


def _oracle_steps(qs, ks, heads, scale, pipe_dtype, acc_dtype):
    raw = ho.RawMaps(acc_dtype)
    for q, k in zip(qs, ks):
        ho.tap(raw, 0, _to_bh(q, heads), _to_bh(k, heads), scale, latent_hw=q.shape[1], pipe_dtype=pipe_dtype)
    return np.stack([v for _, v in raw])          # [kept heads, 77, h, w]


SHAPES = [
    # (batch, heads, side, d)
    (2, 2, 8, 8),          # one partial 128-pixel tile, one k-step with a zero-padded piece
    (2, 3, 16, 40),        # SD-v1.5 head dim 40 (3 k-steps, last half padded)
    (2, 2, 32, 64),        # SDXL head dim
    (2, 1, 24, 80),        # hw = 576: partial tile (4.5 tiles of 128)
    (2, 1, 16, 160),       # SD-v1.5 deepest level
    (1, 4, 16, 64),        # no CFG: keeps heads 2..3 (trace.py:240)
    (4, 2, 16, 16),        # num_images_per_prompt=2 under CFG: keeps batch 2..3
    (2, 2, 10, 12),        # d % 8 != 0 and hw % 8 != 0 -> generic kernel
]


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('mode', ['f16_exact', 'f16_f32acc', 'f32', 'bf16_exact', 'bf16_f32acc'])
@pytest.mark.parametrize('defer', [0, 3])
def test_tap_qk_vs_oracle(shape, mode, defer):
    batch, heads, side, d = shape
    hw = side * side
    steps = 4
    rng = np.random.default_rng(sum(shape) * 7 + len(mode))
    np_dt = np.float32 if mode == 'f32' else ho.BF16 if mode.startswith('bf16') else np.float16
    acc_np = np_dt if mode.endswith('_exact') else np.float32
    scale = d ** -0.5
    qs, ks = zip(*[_qk(rng, batch, heads, hw, d, np_dt) for _ in range(steps)])
    want = _oracle_steps(qs, ks, heads, scale, np_dt, acc_np).astype(np.float64)

    eng = _engine(accumulate='exact' if not mode.endswith('_f32acc') else 'float32', defer_steps=defer)
    for q, k in zip(qs, ks):
        eng.tap_qk(0, _dev(q, np_dt), _dev(k, np_dt), heads, scale, factor=1)
    got = np.stack([v.float().cpu().numpy() for _, v in eng.items()]).astype(np.float64)
    assert got.shape == want.shape
    half_ulp = 2.0 ** -8 if mode.startswith('bf16') else 2.0 ** -11      # of a probability <= 1
    if mode == 'f32':
        tol = 2e-6 * max(1.0, np.abs(want).max())
    elif mode.endswith('_exact'):
        tol = 2 * half_ulp * max(1.0, want.max())         # 1 ulp of the largest running sum
    else:
        tol = steps * half_ulp                            # one flipped probability ulp per step
    err = np.abs(got - want).max()
    assert err <= tol, f'{shape} {mode} defer={defer}: max-abs {err} > {tol}'
    # every step's probabilities sum to one over the tokens
    np.testing.assert_allclose(got.sum(1), steps, atol=steps * 77 * half_ulp)
    eng.close()


@pytest.mark.parametrize('d', [64, 40, 80])
def test_tap_wide_logit_spread(d):
    """The fast softmax takes the exponentials of the logits themselves (no reference point, round 4) and falls back to the true row
    maximum when a pixel's sum leaves [2^-100, 2^100] (1/sum would leave the normal f32 range; further out the exponentials overflow
    / vanish): rows whose logits reach +-36 / 48 / 72 / 96 / 192 next to ordinary rows, and a step whose logits are ALL around -48 /
    -96 / -192 (sums of 2^-63 -- fine -- and of 2^-132 / 0 -- redone), still match the oracle."""
    rng = np.random.default_rng(5)
    heads, side, steps = 2, 16, 4
    hw = side * side
    qs, ks = [], []
    for s in range(steps):
        q, k = _qk(rng, 2, heads, hw, d, np.float16)
        q = q.reshape(2, hw, heads, d)
        k = k.reshape(2, 77, heads, d)
        g = 64.0 / d                                              # keep q.k * scale at +-48 / +-96 for any d
        q[:, ::3] = np.float16(2.0 * np.sqrt(g))
        k[:, 0] = np.float16(-3.0 * np.sqrt(g))                    # token 0 far BELOW the others on those rows
        k[:, 5] = np.float16(3.0 * np.sqrt(g))
        if s == 3:                                                 # every token far below zero: the sum underflows on the wide rows
            k[:] = np.float16(-3.0 * np.sqrt(g)) + (k * np.float16(0.02)).astype(np.float16)
        q[:, 1::3, 1] *= np.float16(2.0)                           # second head: other rows get a 2x wider spread
        q[:, 2::9] = np.float16(0.75 * np.sqrt(g))                 # spread 36: below the switch
        q[:, 5::9] = np.float16(1.5 * np.sqrt(g))                  # spread 72: just above it, exponentials still finite
        qs.append(np.ascontiguousarray(q.reshape(2, hw, heads * d)))
        ks.append(np.ascontiguousarray(k.reshape(2, 77, heads * d)))
    scale = d ** -0.5
    want = _oracle_steps(qs, ks, heads, scale, np.float16, np.float16).astype(np.float64)
    assert np.isfinite(want).all()
    eng = _engine(defer_steps=2)
    for q, k in zip(qs, ks):
        eng.tap_qk(0, _dev(q), _dev(k), heads, scale, factor=1)
    got = np.stack([v.float().cpu().numpy() for _, v in eng.items()]).astype(np.float64)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= 2.0 ** -10 * max(1.0, want.max())
    np.testing.assert_allclose(got.sum(1), steps, atol=steps * 77 * 2.0 ** -11)
    eng.close()


def test_generic_and_mfma_agree(monkeypatch):
    """The baseline (any-shape) kernel and the MFMA kernel implement the same rounding points."""
    monkeypatch.setenv('DAAM_STRICT_EXP', '1')
    rng = np.random.default_rng(7)
    q, k = _qk(rng, 2, 2, 1024, 64, np.float16)
    outs = []
    for force in ('1', '0'):
        monkeypatch.setenv('DAAM_FORCE_GENERIC', force)
        eng = _engine()
        eng.tap_qk(0, torch.from_numpy(q).to(DEV), torch.from_numpy(k).to(DEV), 2, 0.125, factor=2)
        outs.append(np.stack([v.float().cpu().numpy() for _, v in eng.items()]))
        eng.close()
    # different f32 summation order in the dot product: a few logits straddle an fp16 boundary
    diff = np.abs(outs[0] - outs[1])
    assert diff.max() <= 2.0 ** -10
    assert (diff > 0).mean() < 0.02


@pytest.mark.parametrize('steps,d', [(7, 64), (70, 64), (70, 40)])
def test_deferred_equals_immediate_bits(steps, d):
    """Deferring steps changes the launch structure, not one bit of the result -- also when a whole
    50-step generation (or more: 70 steps = a full 64-step launch + 6) runs as one launch, with the fp16
    running sums carried in registers across the steps."""
    rng = np.random.default_rng(11)
    data = [_qk(rng, 2, 2, 256, d, np.float16) for _ in range(steps)]
    res = []
    for defer in (0, 1, 3, 64):
        eng = _engine(defer_steps=defer)
        for q, k in data:
            eng.tap_qk(1, torch.from_numpy(q).to(DEV), torch.from_numpy(k).to(DEV), 2, d ** -0.5, factor=4)
        res.append(torch.stack([v for _, v in eng.items()]).cpu())
        eng.close()
    for r in res[1:]:
        assert torch.equal(res[0], r)


@pytest.mark.parametrize('dt', ['float16', 'float32', 'bfloat16'])
def test_tap_probs_bit_exact(dt):
    rng = np.random.default_rng(3)
    np_dt = ho.BF16 if dt == 'bfloat16' else np.dtype(dt)
    heads, hw, steps = 3, 144, 5                          # hw = 12*12: not a multiple of the 64-pixel tile
    raw = ho.RawMaps(np_dt)
    eng = _engine()
    for s in range(steps):
        q, k = _qk(rng, 2, heads, hw, 16, np_dt)
        probs = ho.attention_probs(_to_bh(q, heads), _to_bh(k, heads), 0.25, np_dt)
        ho.tap(raw, 2, None, None, 0.25, latent_hw=4096, pipe_dtype=np_dt, probs=probs)
        eng.tap_probs(2, _dev(probs, np_dt), factor=5)
    want = np.stack([v for _, v in raw])
    got = torch.stack([v for _, v in eng.items()])
    assert str(got.dtype) == f'torch.{dt}'
    got = got.float().cpu().numpy() if dt == 'bfloat16' else got.cpu().numpy()
    np.testing.assert_array_equal(got, want)
    assert [k for k, _ in eng.items()] == [(5, 2, h) for h in range(heads)]
    eng.close()


@pytest.mark.parametrize('sides', [(64,), (32,), (16, 32, 64), (128, 64), (8,), (24, 48)])
@pytest.mark.parametrize('acc', ['float16', 'float32', 'bfloat16'])
@pytest.mark.parametrize('path', ['default', 'no_pipe', 'no_mfma', 'general'])
def test_finalize_vs_oracle(sides, acc, path, monkeypatch):
    """bicubic (A=-0.75, border-clamped taps) -> clamp -> mean over keys, incl. the x0.5
    down-sample of SDXL-2048 (128 -> 64) and the 96x96 output of 768-px models."""
    # default: the software-pipelined MFMA kernel for fp16 32 -> 64 (other classes beside it on auxiliary streams); no_pipe: round 2's
    # MFMA kernel; no_mfma: the LDS / packed-f32 kernel; general: the any-size kernel
    monkeypatch.setenv('DAAM_NO_MFMA_FINALIZE', '1' if path == 'no_mfma' else '0')
    monkeypatch.setenv('DAAM_NO_PIPE_FINALIZE', '1' if path == 'no_pipe' else '0')
    monkeypatch.setenv('DAAM_FORCE_GENERIC', '1' if path == 'general' else '0')
    rng = np.random.default_rng(len(sides) * 31 + sides[0])
    out_side = 96 if 24 in sides else 64
    heads = 2
    eng = _engine(n_layers=len(sides), out_side=out_side, accumulate='float32' if acc == 'float32' else 'exact')
    np_dt = ho.BF16 if acc == 'bfloat16' else acc
    raw = []
    for layer, side in enumerate(sides):
        # signed planes so that the clamp matters; feed them through the probs path (adds to zero)
        planes = rng.standard_normal((2 * heads, side * side, 77)).astype(np.float32) * 3
        planes = ho.round_bf16(planes) if acc == 'bfloat16' else planes.astype(acc)
        eng.tap_probs(layer, _dev(planes, np_dt), factor=out_side // side if side <= out_side else 0)
        kept = ho.unravel(planes)
        factor = out_side // side if side <= out_side else 0
        raw += [((factor, layer, h), kept[h]) for h in range(heads)]
    lat = out_side * out_side
    for kw in [dict(), dict(head_idx=1), dict(layer_idx=len(sides) - 1), dict(factors=[raw[0][0][0]])]:
        want = ho.global_heat_map(raw, lat, **kw)
        got = eng.global_heat_map(**kw).cpu().numpy()
        tol = 3e-6 * max(1.0, np.abs(want).max())
        np.testing.assert_allclose(got, want, rtol=0, atol=tol, err_msg=f'{sides} {kw}')
    with pytest.raises(LookupError):
        eng.global_heat_map(head_idx=99)
    eng.close()


@pytest.mark.parametrize('n_keys,path', [(4800, 'default'), (8100, 'default'), (8100, 'no_mfma'), (4800, 'no_pipe'), (8100, 'no_pipe')])
def test_finalize_many_x2_keys(n_keys, path, monkeypatch):
    """More 32 x 32 keys than ONE launch of a x2 kernel covers (the MFMA kernel: 31 chunks x 2 key lanes x 64 = 3968; SDXL-1024
    with num_images_per_prompt = 4 has 60 layers x 80 kept heads = 4800): the class is split over several launches, no key
    is dropped.  Key i holds plane set (i mod 81) scaled by 2^-(i div 81 mod 4): bicubic and clamp are positively
    homogeneous, so the expected mean follows from 81 oracle maps -- and dropping ANY subset of keys changes it."""
    monkeypatch.setenv('DAAM_NO_MFMA_FINALIZE', '1' if path == 'no_mfma' else '0')
    monkeypatch.setenv('DAAM_NO_PIPE_FINALIZE', '1' if path == 'no_pipe' else '0')
    rng = np.random.default_rng(n_keys)
    base_n, side = 81, 32
    base = (rng.standard_normal((base_n, side * side, 77)) * 3).astype(np.float16)         # signed: the clamp matters
    per_key = np.stack([ho.global_heat_map([((2, 0, 0), ho.unravel(np.concatenate([base[i:i + 1]] * 2))[0])], 4096)
                        for i in range(base_n)]).astype(np.float64)                        # [81, 77, 64, 64]
    scales = 2.0 ** -((np.arange(n_keys) // base_n) % 4)
    want = np.zeros_like(per_key[0])
    for i in range(n_keys):
        want += scales[i] * per_key[i % base_n]
    want /= n_keys
    eng = _engine(n_layers=1, accumulate='exact')
    bd = torch.from_numpy(base).to(DEV)
    idx = torch.arange(n_keys, device=DEV)
    planes = bd[idx % base_n] * torch.from_numpy(scales.astype(np.float16)).to(DEV)[:, None, None]   # exact: powers of two
    probs = torch.cat([torch.zeros_like(planes), planes])                                  # [2 n_keys, hw, 77], cond half kept
    del planes
    eng.tap_probs(0, probs, factor=2)
    del probs
    got = eng.global_heat_map().cpu().numpy()
    assert np.abs(got - want).max() <= 3e-6 * max(1.0, np.abs(want).max())
    sel = eng.global_heat_map(head_idx=n_keys - 1).cpu().numpy()                           # the last key alone
    np.testing.assert_allclose(sel, scales[-1] * per_key[(n_keys - 1) % base_n], rtol=0, atol=3e-6 * np.abs(per_key).max())
    eng.close()


@pytest.mark.parametrize('fold', ['fold', 'nofold'])
@pytest.mark.parametrize('n_keys', [1, 2, 3, 5, 8, 13, 27, 104, 105, 1000, 1001])
@pytest.mark.parametrize('acc', ['float16', 'bfloat16', 'float32'])
def test_finalize_pipe_key_counts(n_keys, fold, acc, monkeypatch):
    """The software-pipelined x2 kernel walks a pointer table padded with all-zero planes to an even length >= 4 per chunk:
    key counts around every padding / chunking boundary (1 key = 3 padding planes; 13 chunks from 104 keys on; odd shares),
    with a same-size layer (two or five 64 x 64 keys) whose keys ride along in the same kernel (``fold``) or run as their own
    kernel on a second stream (``nofold``).  Key i = plane set (i mod 27) scaled by 2^-(i mod 3).  ``acc`` = dtype of the sums:
    fp16, bf16 (pass 1 on the bf16 MFMA, the tap matrix as two bf16 operands) and f32 (4 KiB planes split into an fp16 hi + lo pair in the
    kernel; folded same-size keys on the VALU) each have their own generated schedule (round 6)."""
    monkeypatch.setenv('DAAM_NO_MFMA_FINALIZE', '0')
    monkeypatch.setenv('DAAM_NO_PIPE_FINALIZE', '0')
    monkeypatch.setenv('DAAM_NO_FOLD_SAME', '1' if fold == 'nofold' else '0')   # same-size keys inside the pipelined kernel / beside it
    rng = np.random.default_rng(1000 + n_keys)
    base_n, side = 27, 32
    np_dt = ho.BF16 if acc == 'bfloat16' else acc
    t_dt = getattr(torch, acc)

    def planes_of(shape):
        x = rng.standard_normal(shape).astype(np.float32) * 3
        return ho.round_bf16(x) if acc == 'bfloat16' else x.astype(acc)
    base = planes_of((base_n, side * side, 77))
    per_key = np.stack([ho.global_heat_map([((2, 0, 0), ho.unravel(np.concatenate([base[i:i + 1]] * 2))[0])], 4096)
                        for i in range(base_n)]).astype(np.float64)
    scales = 2.0 ** -(np.arange(n_keys) % 3)
    n_same = 5 if n_keys >= 8 else 2                                                        # layer 1: 64 x 64 keys
    same = planes_of((2 * n_same, 64 * 64, 77))
    same_maps = [ho.global_heat_map([((1, 1, h), ho.unravel(same)[h])], 4096).astype(np.float64) for h in range(n_same)]
    want = sum(same_maps)
    for i in range(n_keys):
        want = want + scales[i] * per_key[i % base_n]
    want /= n_keys + n_same
    eng = _engine(n_layers=2, accumulate='float32' if acc == 'float32' else 'exact')
    bd = _dev(base, np_dt)
    idx = torch.arange(n_keys, device=DEV)
    planes = bd[idx % base_n] * torch.from_numpy(scales.astype(np.float32)).to(DEV).to(t_dt)[:, None, None]   # exact: powers of two
    eng.tap_probs(0, torch.cat([torch.zeros_like(planes), planes]), factor=2)
    eng.tap_probs(1, _dev(same, np_dt), factor=1)
    for _ in range(2):                                                                     # twice: the table ring advances
        got = eng.global_heat_map().cpu().numpy()
        assert np.abs(got - want).max() <= 3e-6 * max(1.0, np.abs(want).max()), n_keys
    names = eng.last_kernels(1)
    tag = {'float16': 'f16', 'bfloat16': 'bf16', 'float32': 'f32'}[acc]
    folded = fold == 'fold'
    assert f'finalize_up32_pipe_kernel<{tag}' + (' + same-size keys>' if folded else '>') in names, names
    assert ('finalize_same_kernel' in names) == (not folded), names
    only = eng.global_heat_map(factors=[2]).cpu().numpy()                                  # the x2 class alone (no second stream)
    want2 = sum(scales[i] * per_key[i % base_n] for i in range(n_keys)) / n_keys
    assert np.abs(only - want2).max() <= 3e-6 * max(1.0, np.abs(want2).max())
    eng.close()


def test_finalize_prepare_paths(monkeypatch):
    """ABI v5, daam_finalize_prepare: the key tables stay on the device between calls and the output is cleared by the upload
    kernel of the tap launch in front of the finalize -- generation after generation (first call of a selection: tables + zeroing
    at once; later ones: zeroing folded into the tap launch, or a stand-alone zeroing when nothing is pending), across changing
    selections, into RECYCLED output memory (torch hands the previous map's block back: a skipped zeroing would double the map),
    and against the same engine with DAAM_NO_FIN_CACHE=1 (every call uploads + zeroes itself, round 3's behaviour).  Also the raw
    C ABI: an announcement for one buffer followed by a finalize into ANOTHER one must not rely on it."""
    import ctypes
    from daam_amd import _native as nat
    from daam_amd import engine as E
    rng = np.random.default_rng(77)
    shapes = [(2, 32, 64), (2, 64, 64), (3, 16, 40)]                 # heads, side, head_dim
    steps = [[_qk(rng, 2, h, s * s, d, np.float16) for (h, s, d) in shapes] for _ in range(3)]
    dev_steps = [[(_dev(q), _dev(k)) for q, k in st] for st in steps]

    def generation(eng, n):
        eng.clear()
        for t in range(n):
            for layer, ((h, s, d), (q, k)) in enumerate(zip(shapes, dev_steps[t % 3])):
                eng.tap_qk(layer, q, k, h, d ** -0.5, 64 // s)

    selections = [dict(), dict(factors=[2]), dict(head_idx=1), dict(), dict(), dict(layer_idx=0), dict()]
    results = {}
    for cache in ('1', '0'):
        monkeypatch.setenv('DAAM_NO_FIN_CACHE', '1' if cache == '0' else '0')
        E.release_parked_contexts()
        eng = _engine(n_layers=len(shapes), defer_steps=8)
        got = []
        for g, kw in enumerate(selections):
            generation(eng, 2 + g % 2)
            a = eng.global_heat_map(**kw)                            # taps pending: prepare -> flush (folds) -> finalize
            got.append(a.clone())
            del a                                                   # its block goes back to the allocator ...
            b = eng.global_heat_map(**kw)                            # ... and is handed out again; nothing pending now
            assert torch.allclose(b, got[-1], rtol=0, atol=2e-6), (cache, g, kw)
            del b
        results[cache] = got
        if cache == '1':
            # raw ABI: announce buffer A, finalize into B (filled with garbage): B must come out right; then the announced form
            lib, stream = eng.lib, eng.stream
            A = torch.full((77, 64, 64), 7.0, device=DEV)
            B = torch.full((77, 64, 64), -3.0, device=DEV)
            nat.check(lib.daam_finalize_prepare(eng.ctx, None, 0, A.data_ptr(), stream))
            nat.check(lib.daam_finalize(eng.ctx, None, 0, B.data_ptr(), stream))
            assert torch.allclose(B, got[-1], rtol=0, atol=2e-6)
            A.fill_(9.0)
            nat.check(lib.daam_finalize_prepare(eng.ctx, None, 0, A.data_ptr(), stream))
            nat.check(lib.daam_finalize(eng.ctx, None, 0, A.data_ptr(), stream))
            assert torch.allclose(A, got[-1], rtol=0, atol=2e-6)
            # an announcement is one-shot: the next finalize into the same buffer zeroes it itself
            nat.check(lib.daam_finalize(eng.ctx, None, 0, A.data_ptr(), stream))
            assert torch.allclose(A, got[-1], rtol=0, atol=2e-6)
        eng.close()
    for g, (a, b) in enumerate(zip(results['1'], results['0'])):
        assert float(b.abs().max()) > 0
        assert torch.allclose(a, b, rtol=0, atol=2e-6), (g, float((a - b).abs().max()))
    E.release_parked_contexts()


@pytest.mark.parametrize('sides', [(32, 64), (16, 32, 64), (128, 64), (24, 48)])
@pytest.mark.parametrize('acc', ['float16', 'float32', 'bfloat16'])
def test_finalize_n_rows(sides, acc):
    """ABI v6: ``daam_finalize(..., n_rows, ...)`` computes the token rows [0, n_rows) only (the crop of daam/trace.py:127 applied before the
    work).  For every class kernel: rows [0, n_rows) equal the oracle's / the 77-row call's, rows >= n_rows of a 77-row buffer keep
    the sentinel they held, the announced form (daam_finalize_prepare with the same / with another row count) agrees, and the engine
    returns an [n_rows, x, x] tensor."""
    import ctypes
    from daam_amd import _native as nat
    rng = np.random.default_rng(sides[0] * 7 + len(sides))
    out_side = 96 if 24 in sides else 64
    heads = 3
    eng = _engine(n_layers=len(sides), out_side=out_side, accumulate='float32' if acc == 'float32' else 'exact')
    np_dt = ho.BF16 if acc == 'bfloat16' else acc
    raw = []
    for layer, side in enumerate(sides):
        planes = rng.standard_normal((2 * heads, side * side, 77)).astype(np.float32) * 3
        planes = ho.round_bf16(planes) if acc == 'bfloat16' else planes.astype(acc)
        factor = out_side // side if side <= out_side else 0
        eng.tap_probs(layer, _dev(planes, np_dt), factor=factor)
        kept = ho.unravel(planes)
        raw += [((factor, layer, h), kept[h]) for h in range(heads)]
    want = ho.global_heat_map(raw, out_side * out_side)
    tol = 3e-6 * max(1.0, np.abs(want).max())
    full = eng.global_heat_map()
    np.testing.assert_allclose(full.cpu().numpy(), want, rtol=0, atol=tol)
    lib, stream = eng.lib, eng.stream
    for n_rows in (1, 4, 12, 76, 77):
        got = eng.global_heat_map(n_rows=n_rows)
        assert tuple(got.shape) == (n_rows, out_side, out_side)
        np.testing.assert_allclose(got.cpu().numpy(), want[:n_rows], rtol=0, atol=tol, err_msg=f'{sides} {acc} n_rows={n_rows}')
        # raw ABI into a 77-row buffer of sentinels: plain call, announced call, announcement for ANOTHER row count
        for announce in (None, n_rows, 77 if n_rows != 77 else 5):
            buf = torch.full((77, out_side, out_side), -123.0, device=DEV)
            if announce is not None:
                nat.check(lib.daam_finalize_prepare(eng.ctx, None, announce, buf.data_ptr(), stream))
                if announce != n_rows:
                    torch.cuda.synchronize()
                    buf.fill_(-123.0)                      # what the announcement cleared is not this call's business
            nat.check(lib.daam_finalize(eng.ctx, None, n_rows, buf.data_ptr(), stream))
            np.testing.assert_allclose(buf[:n_rows].cpu().numpy(), want[:n_rows], rtol=0, atol=tol,
                                       err_msg=f'{sides} {acc} n_rows={n_rows} announce={announce}')
            assert bool((buf[n_rows:] == -123.0).all()), (sides, acc, n_rows, announce)
    # out of range = every row
    buf = torch.full((77, out_side, out_side), -123.0, device=DEV)
    nat.check(lib.daam_finalize(eng.ctx, None, 500, buf.data_ptr(), stream))
    np.testing.assert_allclose(buf.cpu().numpy(), want, rtol=0, atol=tol)
    assert eng.last_kernels(1) != ''
    eng.close()


def test_raw_abi_strides_past_32_bit_offsets_fall_back():
    """A raw C-ABI caller whose Q is a view into a large fused buffer (batch stride 2^30 elements = 2 GiB in fp16): the
    specialised tap kernels address Q / K with 32-bit byte offsets, so such a call must take the any-shape kernel instead of
    wrapping (round-3 advisor finding) -- same sums as the same data in standard layout, immediate and deferred."""
    import ctypes
    from daam_amd import _native as nat
    heads, side, d, batch = 2, 16, 64, 2
    hw, c = side * side, heads * d
    g = torch.Generator(device=DEV).manual_seed(9)
    q = torch.randn(batch, hw, c, generator=g, device=DEV, dtype=torch.float16)
    k = torch.randn(batch, 77, c, generator=g, device=DEV, dtype=torch.float16)
    stride_b = 1 << 30
    big = torch.zeros(stride_b + hw * c, device=DEV, dtype=torch.float16)          # 2 GiB + one batch element
    big[:hw * c] = q[0].reshape(-1)
    big[stride_b:] = q[1].reshape(-1)
    scale = d ** -0.5

    ref = _engine(n_layers=1, defer_steps=0)
    ref.tap_qk(0, q, k, heads, scale, 4)
    want = {key: v.clone() for key, v in ref.items()}
    ref.close()

    for deferred in (False, True):
        eng = _engine(n_layers=1, defer_steps=0)
        eng._require_device(big)                                                   # binds the engine to the device (tap_qk does this)
        eng._ensure_ctx(torch.float16)
        eng._ensure_layer(0, heads, side, 4)                                       # kept batch*heads entries = BH - BH/2 = heads
        eng._touch(0)
        desc = nat.QKDesc(in_dtype=0, batch=batch, heads=heads, hw=hw, tokens=77, head_dim=d, round_logits=1, scale=float(scale),
                          q_stride_b=stride_b, q_stride_h=d, q_stride_p=c, k_stride_b=77 * c, k_stride_h=d, k_stride_t=c)
        if deferred:
            nat.check(eng.lib.daam_tap_qk_enqueue(eng.ctx, 0, big.data_ptr(), k.data_ptr(), ctypes.byref(desc)))
            nat.check(eng.lib.daam_tap_flush(eng.ctx, eng.stream))
        else:
            nat.check(eng.lib.daam_tap_qk(eng.ctx, 0, big.data_ptr(), k.data_ptr(), ctypes.byref(desc), eng.stream))
        torch.cuda.synchronize()
        grid, block, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        nat.check(eng.lib.daam_last_launch(eng.ctx, 0, ctypes.byref(grid), ctypes.byref(block), ctypes.byref(lds)))
        got = {key: v.clone() for key, v in eng.items()}
        assert list(got) == list(want)
        for key in want:
            a, b = got[key].float(), want[key].float()
            assert float(b.abs().max()) > 0
            # another kernel (f32 FMA dot products): the tolerance of the kernel-path tests
            assert float((a - b).abs().max()) <= 2.0 ** -10 * max(1.0, float(b.max())), (deferred, key)
        eng.close()
    del big


def test_profile_history_times_every_launch_of_a_region():
    """daam_profile_enable(ctx, 2) + daam_profile_history: one HIP-event pair per tap launch / finalize call out of a ring, read back
    after the region (what bench.py's roofline.ms_per_launch is made of) -- counts, order, capacity clipping, and plausible durations."""
    import ctypes
    from daam_amd import _native as nat
    rng = np.random.default_rng(3)
    heads, side, d = 2, 32, 64
    q, k = _qk(rng, 2, heads, side * side, d, np.float16)
    qd, kd = _dev(q), _dev(k)
    eng = _engine(n_layers=1, defer_steps=8)
    eng.tap_qk(0, qd, kd, heads, d ** -0.5, 2)
    eng.global_heat_map()                                            # creates the context, warms up
    nat.check(eng.lib.daam_profile_enable(eng.ctx, 2))
    steps = [1, 3, 5, 2]                                             # steps per generation: launch durations grow with them
    for n in steps:
        eng.clear()
        for _ in range(n):
            eng.tap_qk(0, qd, kd, heads, d ** -0.5, 2)
        eng.global_heat_map()

    def history(which, cap):
        buf = (ctypes.c_float * 16)()
        got = ctypes.c_int()
        nat.check(eng.lib.daam_profile_history(eng.ctx, which, buf, cap, ctypes.byref(got)))
        return [buf[i] for i in range(got.value)]
    tap, fin = history(0, 16), history(1, 16)
    assert len(tap) == len(steps) and len(fin) == len(steps)
    assert all(0.0 < t < 50.0 for t in tap + fin), (tap, fin)
    assert history(0, 2) == tap[-2:]                                 # clipped to the newest launches, oldest first
    # daam_profile_last_ms in ring mode reads the NEWEST ring slot (not a stale pair of an earlier mode-1 run)
    last = ctypes.c_float()
    nat.check(eng.lib.daam_profile_last_ms(eng.ctx, 0, ctypes.byref(last)))
    assert last.value == tap[-1]
    nat.check(eng.lib.daam_profile_last_ms(eng.ctx, 1, ctypes.byref(last)))
    assert last.value == fin[-1]
    nat.check(eng.lib.daam_profile_enable(eng.ctx, 2))               # re-arming starts a new region
    assert history(0, 16) == []
    assert eng.lib.daam_profile_last_ms(eng.ctx, 0, ctypes.byref(last)) != 0     # nothing launched in the new region yet: an error, not a stale time
    nat.check(eng.lib.daam_profile_enable(eng.ctx, 0))
    eng.close()


def test_failed_announcement_still_launches_and_drops_the_recorded_calls():
    """``flush(_before_launch=...)``: the recorded calls are in the library's hands when the announcement runs.  If it raises, the launch
    that consumes them must still follow (the Python references to their Q / K are dropped right after: entries left pending would be
    read from freed memory by the next launch) and the error must reach the caller.  (Advisor, round 4.)"""
    import ctypes
    from daam_amd import _native as nat
    heads, hw, d = 2, 256, 64
    g = torch.Generator(device=DEV).manual_seed(3)
    q = torch.randn(2, hw, heads * d, generator=g, device=DEV, dtype=torch.float16)
    k = torch.randn(2, 77, heads * d, generator=g, device=DEV, dtype=torch.float16)
    ref = _engine(defer_steps=8)
    ref.tap_qk(0, q, k, heads, d ** -0.5, factor=4)
    ref.flush()
    want = torch.stack([v.clone() for _, v in ref.items()])
    ref.close()
    eng = _engine(defer_steps=8)
    eng.tap_qk(0, q, k, heads, d ** -0.5, factor=4)

    def boom(stream):
        raise RuntimeError('announcement failed')
    with pytest.raises(RuntimeError, match='announcement failed'):
        eng.flush(_before_launch=boom)
    n_calls, max_steps = ctypes.c_int(-1), ctypes.c_int(-1)
    nat.check(eng.lib.daam_tap_pending(eng.ctx, ctypes.byref(n_calls), ctypes.byref(max_steps)))
    assert n_calls.value == 0 and eng.pending_taps == 0            # nothing left behind on either side
    got = torch.stack([v.clone() for _, v in eng.items()])
    assert torch.equal(got, want)                                    # the launch ran: the step is in the sums
    eng.tap_qk(0, q, k, heads, d ** -0.5, factor=4)                # and the engine goes on working
    eng.flush()
    torch.cuda.synchronize()
    eng.close()


def test_views_survive_clear_and_next_generation():
    """``all_heat_maps`` hands out views of the live sums; like the reference's tensors (heatmap.py:170-172: clear() drops the
    dict, tensors handed out before live on) they must keep their values through clear() AND through the next
    generation's finalize -- also for a layer the next generation does not tap (the native context kept the
    old buffer and zeroed it)."""
    rng = np.random.default_rng(5)
    eng = _engine(n_layers=2, accumulate='exact', defer_steps=4)
    q, k = _qk(rng, 2, 2, 32 * 32, 64, np.float16)
    qd, kd = _dev(q), _dev(k)
    for layer in (0, 1):
        eng.tap_qk(layer, qd, kd, 2, 0.125, factor=2)
    views = {key: v for key, v in eng.items()}
    before = {key: v.clone() for key, v in views.items()}
    assert len(views) == 4 and all(float(v.sum()) > 0 for v in views.values())
    eng.clear()
    eng.tap_qk(0, qd, kd, 2, 0.125, factor=2)                  # generation 2 taps layer 0 only
    assert [key for key, _ in eng.items()] == [(2, 0, 0), (2, 0, 1)]
    gm = eng.global_heat_map()
    torch.cuda.synchronize()
    for key, v in views.items():
        assert torch.equal(v, before[key]), key                # neither overwritten nor zeroed
    assert float(gm.sum()) > 0
    eng.close()


def test_normalize_and_word_maps():
    rng = np.random.default_rng(5)
    maps = np.abs(rng.standard_normal((9, 64, 64))).astype(np.float32)
    from daam_amd import engine as E
    eng = _engine()
    t = torch.from_numpy(maps.copy()).to(DEV)
    eng._require_device(t)
    got = eng.normalize_(t).cpu().numpy()
    want = maps / (maps[1:-1].sum(0, keepdims=True) + np.float32(1e-6))
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)
    gm = torch.from_numpy(maps).to(DEV)
    wm = E.word_heat_map(gm, [2, 3, 5])
    np.testing.assert_allclose(wm.cpu().numpy(), ho.word_heat_map(maps, [2, 3, 5]), rtol=1e-6, atol=1e-7)
    for size in (64, 128, 512):
        for kw in (dict(), dict(absolute=True), dict(threshold=0.4)):
            got = E.expand_word_map(wm, size, size, **kw).cpu().numpy()
            want = ho.expand_as(wm.cpu().numpy(), size, **kw)
            if 'threshold' in kw:
                assert (got != want).mean() < 1e-4          # values within 1e-6 of the threshold may flip
            else:
                np.testing.assert_allclose(got, want, rtol=0, atol=2e-5)
    eng.close()


# ---------------------------------------------------------------------------------------------
# end to end through the reference's API against the golden vectors of the unmodified reference
# ---------------------------------------------------------------------------------------------
def _global_tol(meta):
    # north_star: <= 1e-3 max-abs in fp16; bf16 carries 3 bits less
    return {'float32': 2e-6, 'float16': 1e-3, 'bfloat16': 8e-3}[meta['dtype']]


def _out_tol(meta):
    # processor outputs (trace.py:296-304), relative to the largest |value| of the compared rows: fp32 = summation
    # order of the GEMMs; fp16 / bf16 = the fused route keeps the probabilities in f32 up to P.V where the reference
    # rounds them to the pipeline dtype first, plus the output rounding
    return {'float32': 1e-5, 'float16': 2e-3, 'bfloat16': 1.6e-2}[meta['dtype']]


def _check_processor_outputs(pipe, z, meta, sums_key='out_sums', rows=True):
    """``hidden_states`` returned by every hooked / un-hooked cross-attention call of the last step against what the
    REFERENCE's processor returned (fixtures: per-call sum of squares + the first rows of the conditional batch)."""
    outs = pipe.last_outputs
    assert len(outs) == len(z[sums_key])
    rel = _out_tol(meta)
    for i, o in enumerate(outs):
        sq = float((o.double() ** 2).sum())
        assert abs(sq - z[sums_key][i, 1]) <= 4 * rel * z[sums_key][i, 1], f'call {i}: sum of squares'
        if rows:
            want = z[f'out_rows_{i}']
            got = o[-1, :OUT_SAMPLE_ROWS].float().cpu().numpy()
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= rel * max(float(np.abs(want).max()), 1e-6), f'call {i}: rows'


@pytest.mark.parametrize('tap,defer', [('qk', 0), ('qk', 2), ('qk', 50), ('probs', 0)])
def test_trace_api_matches_reference_golden(golden_case, tap, defer, tmp_path):
    import daam_amd
    name, z, meta = golden_case
    pipe = golden_pipe(meta, device=DEV)
    pipe.keep_outputs = True
    # the save_heads cases: the locator also returns the mid block and every call goes through the materialised route
    trace_kw = dict(save_heads=True, data_dir=str(tmp_path)) if meta.get('heads') else {}
    with daam_amd.trace(pipe, tap=tap, defer_steps=defer, **trace_kw) as tc:
        out = pipe(meta['prompt'], num_inference_steps=meta['steps'], callback=tc.time_callback)
        _check_processor_outputs(pipe, z, meta)
        items = list(tc.all_heat_maps)
        keys = np.asarray([k for k, _ in items], dtype=np.int32)
        np.testing.assert_array_equal(keys, z['keys'])              # same keys, same first-update order
        assert str(items[0][1].dtype) == str(z['raw_dtype'])
        sums = np.asarray([float(v.double().sum()) for _, v in items])
        np.testing.assert_allclose(sums, z['key_sum'], rtol={'float32': 1e-5, 'float16': 2e-4, 'bfloat16': 2e-3}[meta['dtype']])
        # every (key, token) plane of the reference's run, by two checksums (sum and position-weighted sum)
        rt = {'float32': 1e-5, 'float16': 1e-3, 'bfloat16': 8e-3}[meta['dtype']]
        ps = np.stack([v.double().sum((1, 2)).cpu().numpy() for _, v in items])
        pw = np.stack([(v.double() * torch.arange(1, v.shape[1] * v.shape[2] + 1, dtype=torch.float64, device=v.device)
                        .view(1, v.shape[1], v.shape[2])).sum((1, 2)).cpu().numpy() for _, v in items])
        np.testing.assert_allclose(ps, z['plane_sum'], rtol=rt, atol=rt)
        np.testing.assert_allclose(pw, z['plane_wsum'], rtol=rt, atol=rt * 1e3)
        for sid in z['raw_sample_ids']:
            got = items[int(sid)][1][SAMPLE_TOKENS].float().cpu().numpy()
            want = z[f'raw_{int(sid)}']
            ulp = {'float16': 2.0 ** -10, 'bfloat16': 2.0 ** -7}.get(meta['dtype'])
            tol = 2e-6 if ulp is None else ulp * max(1.0, float(want.max()))
            np.testing.assert_allclose(got, want, rtol=0, atol=tol)
        for vn, kw in json.loads(str(z['variants'])).items():
            got = tc.compute_global_heat_map(**kw).heat_maps
            want = z[f'global_{vn}']
            assert got.dtype == torch.float32 and tuple(got.shape) == want.shape
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=0,
                                       atol=_global_tol(meta) * max(1.0, float(np.abs(want).max())),
                                       err_msg=f'{name}:{vn}')
        ghm = tc.compute_global_heat_map()
        whm = ghm.compute_word_heat_map(str(z['word']))
        np.testing.assert_allclose(whm.heatmap.cpu().numpy(), z['word_map'], rtol=0, atol=_global_tol(meta))

        class _Img:
            size = (128, 128)
        # min-max normalisation divides by the range of the word map: scale the tolerance with it
        span = float(z['word_map'].max() - z['word_map'].min())
        np.testing.assert_allclose(whm.expand_as(_Img()).numpy(), z['word_expand_128'], rtol=0,
                                   atol=max(2e-5, 4 * _global_tol(meta) / max(span, 1e-6)) if meta['dtype'] != 'float32'
                                   else 5e-5)
        got_abs = whm.expand_as(_Img(), absolute=True).numpy()
        if not np.allclose(got_abs, z['word_expand_128_abs'], rtol=0, atol=max(2e-5, _global_tol(meta))):
            # seen ONCE (round 6, four test processes on one GPU): 64 consecutive elements off.  Keep what is needed to tell a lost write from a late one.
            os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
            np.savez(os.path.join(ROOT, 'gpurun_out', f'expand_mismatch_{name}_{os.getpid()}.npz'), got=got_abs, again=whm.expand_as(_Img(), absolute=True).numpy(),
                     want=z['word_expand_128_abs'], norm=z['word_expand_128'], word=whm.heatmap.cpu().numpy())
        np.testing.assert_allclose(got_abs, z['word_expand_128_abs'], rtol=0, atol=max(2e-5, _global_tol(meta)))
        assert tc.layer_names == json.loads(str(z['layer_names']))
        assert tc.last_prompt == str(z['last_prompt'])
        assert str(tc.last_image) == str(z['last_image'])
        assert int(tc.time_idx) == int(z['time_idx'])
        if meta.get('heads'):
            assert tc._gen_idx == int(z['heads_n_files'])
    assert out.images
    if not meta.get('heads'):
        return
    # ---- heads cache: one ``{gen_idx}.pt`` per processor call with the probabilities [B*H, hw, 77] (trace.py:246-247)
    n_files = int(z['heads_n_files'])
    dtypes = json.loads(str(z['heads_dtypes']))
    rtol = {'float32': 1e-5, 'float16': 1e-3}[meta['dtype']]
    for i in range(n_files):
        t = torch.load(tmp_path / f'{i}.pt')
        assert list(t.shape) == z['heads_shapes'][i].tolist() and str(t.dtype) == dtypes[i]
        np.testing.assert_allclose([float(t.double().sum()), float((t.double() ** 2).sum())], z['heads_stats'][i], rtol=rtol)
    assert not (tmp_path / f'{n_files}.pt').exists()
    # ---- replay (trace.py:281-282): another pipeline (other hidden states, other V) under load_heads=True reads the
    # saved probabilities back -> the same maps, and attention outputs = saved probabilities x ITS values
    pipe2 = golden_pipe(meta, device=DEV, seed_offset=100)
    pipe2.keep_outputs = True
    with daam_amd.trace(pipe2, load_heads=True, data_dir=str(tmp_path), tap=tap, defer_steps=defer) as tc2:
        pipe2(meta['prompt'], num_inference_steps=meta['steps'])
        got = tc2.compute_global_heat_map().heat_maps.cpu().numpy()
        np.testing.assert_allclose(got, z['global_replay'], rtol=0,
                                   atol=_global_tol(meta) * max(1.0, float(np.abs(z['global_replay']).max())))
        _check_processor_outputs(pipe2, z, meta, sums_key='replay_out_sums', rows=False)


@pytest.mark.parametrize('env', [dict(DAAM_STRICT_EXP='1'), dict(DAAM_NO_D64='1'), dict(DAAM_NO_D64='1', DAAM_STRICT_EXP='1'),
                                 dict(DAAM_FORCE_GENERIC='1'), dict(DAAM_NO_PIPE_FINALIZE='1'),
                                 dict(DAAM_NO_PIPE_FINALIZE='1', DAAM_NO_PAIRED_FINALIZE='1'), dict(DAAM_NO_SIDE_STREAM='1'),
                                 dict(DAAM_NO_FOLD_SAME='1')])
def test_optional_kernel_paths_keep_parity(env, monkeypatch):
    """The opt-in / fallback kernel variants (compensated-exp softmax, the 32x32-tile MFMA kernel instead of the
    head_dim-64 one, the any-shape kernels) stay within the same tolerances on an SDXL-shaped fp16 case (head_dim 64)."""
    import daam_amd
    from daam_amd import engine as E
    E.release_parked_contexts()                                  # the switches are read when a context is created
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    from oracle import fake_diffusers as fd
    steps, prompt = 4, 'a photo of a monkey'
    kw = dict(dtype=torch.float16, batch=2, seed=21, mini=True, identity_proj=True, dim_head=64, heads_scale=0.1,
              tblocks_cap=1)
    pipe = fd.make_pipe('sdxl', device=DEV, **kw)
    with daam_amd.trace(pipe, defer_steps=3) as tc:
        pipe(prompt, num_inference_steps=steps)
        got_raw = {k: v.float().cpu().numpy() for k, v in tc.all_heat_maps}
        got = tc.compute_global_heat_map().heat_maps.cpu().numpy()
    cpu_pipe = fd.make_pipe('sdxl', device='cpu', **kw)
    raw = ho.replay_generation(cpu_pipe, steps, torch.float16)
    assert [k for k, _ in raw] == list(got_raw)
    for k, v in raw:
        assert np.abs(got_raw[k] - v.astype(np.float32)).max() <= 2.0 ** -10 * max(1.0, float(v.max()))
    lat = ho.latent_hw_for(cpu_pipe.unet.config.sample_size, cpu_pipe.vae_scale_factor)
    want = ho.global_heat_map(raw, lat, n_rows=len(cpu_pipe.tokenizer.tokenize(prompt)) + 2)
    assert np.abs(got - want).max() <= 1e-3


def test_trace_errors_and_reset():
    import daam_amd
    z, meta = load_golden('sd15_f16')
    pipe = golden_pipe(meta, device=DEV)
    tc = daam_amd.trace(pipe)
    with pytest.raises(RuntimeError, match='Module is not hooked'):
        tc.unhook()
    with tc:
        with pytest.raises(RuntimeError, match='Already hooked module'):
            tc.hook()
        with pytest.raises(RuntimeError, match='Did you forget'):
            tc.compute_global_heat_map()
        with pytest.raises(ValueError, match='Only single prompt'):
            pipe(['a', 'b'])
        pipe('a dog', num_inference_steps=2)
        a = tc.compute_global_heat_map().heat_maps.clone()
        with pytest.raises(RuntimeError, match='given parameters'):
            tc.compute_global_heat_map(layer_idx=999)
        pipe('a dog', num_inference_steps=2)                         # check_inputs clears the sums (trace.py:179)
        b = tc.compute_global_heat_map().heat_maps
        assert torch.allclose(a, b, rtol=0, atol=1e-6)                # f32 atomics: order may differ
        with pytest.raises(ValueError, match='not found in prompt'):
            tc.compute_global_heat_map().compute_word_heat_map('cat')
    assert type(pipe.unet.execution_order()[0].module.processor).__name__ == 'DefaultProcessor'


def test_cpu_tensors_fail_loudly():
    import daam_amd
    z, meta = load_golden('sd15_f32')
    pipe = golden_pipe(meta, device='cpu')
    with daam_amd.trace(pipe) as tc:
        with pytest.raises(RuntimeError, match='no CPU fallback'):
            pipe('a dog', num_inference_steps=1)


# ---------------------------------------------------------------------------------------------
# full-size (BASELINE.json configs) properties that do not need the oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize('heads,side,d', [(10, 64, 64), (20, 32, 64), (8, 64, 40), (5, 128, 64)])
def test_full_size_layer_properties(heads, side, d):
    """SDXL / SD-v1.5 / SDXL-2048 layer shapes: (i) the token sums of the running sums equal the
    number of steps at every pixel; (ii) deferral does not change a bit; (iii) an all-equal-keys
    input gives the uniform map 1/77; (iv) finalize of one key at the output resolution is a
    clamp-copy of the sums."""
    hw = side * side
    g = torch.Generator(device='cpu').manual_seed(heads * 1000 + side)
    steps = 3
    qs = [torch.randn(2, hw, heads * d, generator=g).half().to(DEV) for _ in range(steps)]
    k = torch.randn(2, 77, heads * d, generator=g).half()
    k[:, 0] *= 3
    k = k.to(DEV)
    res = []
    for defer in (0, 8):
        eng = _engine(accumulate='float32', defer_steps=defer)
        for q in qs:
            eng.tap_qk(0, q, k, heads, d ** -0.5, factor=max(0, 64 // side))
        acc = torch.stack([v for _, v in eng.items()])
        res.append(acc.clone())
        tok_sum = acc.sum(1)
        assert (tok_sum - steps).abs().max().item() <= steps * 77 * 2.0 ** -12
        if side == 64:
            gm = eng.global_heat_map(head_idx=1)
            assert torch.allclose(gm, acc[1].clamp(min=0), atol=1e-6)
        eng.close()
    assert torch.equal(res[0], res[1]), f'{(res[0] != res[1]).sum().item()} elements differ, max {(res[0] - res[1]).abs().max().item()}'
    eng = _engine()
    kc = k.clone()
    kc[:] = kc[:, :1]                                            # all keys identical -> uniform attention
    eng.tap_qk(0, qs[0], kc, heads, d ** -0.5, factor=1)
    acc = torch.stack([v for _, v in eng.items()]).float()
    import os
    exact = os.environ.get('DAAM_STRICT_EXP', '0') == '1'      # the default (fast) softmax may be one fp16 ulp off here
    assert (acc - float(np.float16(1.0 / 77))).abs().max().item() <= (0.0 if exact else 2.0 ** -17)
    eng.close()


@pytest.mark.parametrize('side', [24, 48])
@pytest.mark.parametrize('mode', ['f16_exact', 'bf16_exact', 'f16_f32acc'])
def test_eight_wave_partial_tiles_50_deferred_steps_vs_oracle(side, mode, monkeypatch):
    """The eight-wave head_dim-64 form (256-pixel tiles; round 5: for bf16 / f32 sums as well) on layers whose last tile is PARTIAL --
    hw = 576 (SD-2.x at 768 px: 2.25 tiles) and 2304 (9 tiles) -- through a 50-step deferred launch, against the numpy oracle; and
    bit-identical to the four-wave form (DAAM_TAP_W8=0).  Reference: daam/trace.py:233,240 (unravel), heatmap.py:153-156 (the running sum)."""
    import ctypes
    from daam_amd import _native as nat
    from daam_amd import engine as E
    heads, d, steps, hw = 2, 64, 50, side * side
    rng = np.random.default_rng(side * 31 + len(mode))
    np_dt = ho.BF16 if mode.startswith('bf16') else np.float16
    acc_np = np_dt if mode.endswith('_exact') else np.float32
    scale = d ** -0.5
    qs, ks = zip(*[_qk(rng, 2, heads, hw, d, np_dt) for _ in range(steps)])
    want = _oracle_steps(qs, ks, heads, scale, np_dt, acc_np).astype(np.float64)
    got = {}
    for tag, env in (('default', {}), ('four_waves', dict(DAAM_TAP_W8='0'))):
        E.release_parked_contexts()                          # the switches are read when a native context is created
        for var in ('DAAM_TAP_W8',):
            monkeypatch.delenv(var, raising=False)
        for var, val in env.items():
            monkeypatch.setenv(var, val)
        eng = _engine(accumulate='exact' if mode.endswith('_exact') else 'float32', defer_steps=64)
        for q, k in zip(qs, ks):
            eng.tap_qk(0, _dev(q, np_dt), _dev(k, np_dt), heads, scale, factor=1)
        got[tag] = torch.stack([v.float() for _, v in eng.items()]).cpu()
        assert eng.last_flush()['max_steps'] == steps
        grid, block, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        nat.check(eng.lib.daam_last_launch(eng.ctx, 0, ctypes.byref(grid), ctypes.byref(block), ctypes.byref(lds)))
        assert block.value == (256 if tag == 'four_waves' else 512), (tag, block.value)
        if tag != 'four_waves':
            assert grid.value >= heads * -(-hw // 256)       # 256-pixel tiles: 3 / 9 per head (the grid is rounded up to 8 XCDs)
            assert (lds.value >= 69 * 1024) == mode.endswith('_f32acc'), (tag, lds.value)
        eng.close()
    E.release_parked_contexts()
    assert torch.equal(got['default'], got['four_waves'])
    g = got['default'].numpy().astype(np.float64)
    assert g.shape == want.shape
    # tolerance of the 50-step full-size tests (tests/test_gpu_integration.py): the implementations round at the same points but sum
    # q.k in another order, which flips the rounding of a logit now and then; an fp16 / bf16 running sum then differs by an ulp of
    # ITS magnitude at most a few times over 50 steps
    ulp = 2.0 ** -8 if mode.startswith('bf16') else 2.0 ** -11
    if mode.endswith('_exact'):
        tol = 2.0 ** -6 * np.abs(want) + 2 * ulp * max(1.0, want.max()) if not mode.startswith('bf16') else 2.0 ** -4 * np.abs(want) + 2 * ulp * max(1.0, want.max())
    else:
        tol = steps * ulp
    bad = np.abs(g - want) > tol
    assert not bad.any(), f'{mode} side {side}: {bad.sum()} elements beyond tolerance, max-abs {np.abs(g - want).max()}'
    np.testing.assert_allclose(g.sum(1), steps, atol=steps * 77 * ulp)      # every step's probabilities sum to one over the tokens


# ---------------------------------------------------------------------------------------------
# property tests (hypothesis): random layer shapes / step counts / launch structures
# ---------------------------------------------------------------------------------------------
from hypothesis import HealthCheck, given, settings, strategies as st


@settings(max_examples=40, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(heads=st.integers(1, 4), side=st.sampled_from([4, 8, 12, 16, 24]), d=st.sampled_from([8, 16, 24, 40, 64, 80]),
       batch=st.sampled_from([1, 2, 4]), steps=st.integers(1, 5), defer=st.sampled_from([0, 1, 2, 7]),
       mode=st.sampled_from(['f16_exact', 'f16_f32acc', 'f32', 'bf16_exact', 'bf16_f32acc']), seed=st.integers(0, 2 ** 16))
def test_tap_property(heads, side, d, batch, steps, defer, mode, seed):
    """Any (heads, map size, head_dim, CFG batch, steps, deferral, pipeline dtype) combination matches the oracle
    within the stated tolerance and keeps the token sums at `steps`."""
    hw = side * side
    rng = np.random.default_rng(seed)
    np_dt = np.float32 if mode == 'f32' else ho.BF16 if mode.startswith('bf16') else np.float16
    acc_np = np_dt if mode.endswith('_exact') else np.float32
    scale = d ** -0.5
    qs, ks = zip(*[_qk(rng, batch, heads, hw, d, np_dt) for _ in range(steps)])
    want = _oracle_steps(qs, ks, heads, scale, np_dt, acc_np).astype(np.float64)
    eng = _engine(accumulate='float32' if mode.endswith('_f32acc') else 'exact', defer_steps=defer)
    for q, k in zip(qs, ks):
        eng.tap_qk(0, _dev(q, np_dt), _dev(k, np_dt), heads, scale, factor=1)
    got = np.stack([v.float().cpu().numpy() for _, v in eng.items()]).astype(np.float64)
    eng.close()
    assert got.shape == want.shape
    half_ulp = 2.0 ** -8 if mode.startswith('bf16') else 2.0 ** -11
    if mode == 'f32':
        tol = 2e-6 * max(1.0, np.abs(want).max())
    elif mode.endswith('_exact'):
        tol = 2 * half_ulp * max(1.0, want.max())
    else:
        tol = steps * half_ulp
    assert np.abs(got - want).max() <= tol
    np.testing.assert_allclose(got.sum(1), steps, atol=steps * 77 * half_ulp)


@settings(max_examples=80, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
@given(sides=st.lists(st.sampled_from([8, 16, 32, 32, 64, 128]), min_size=1, max_size=4), acc=st.sampled_from(['float16', 'float32', 'bfloat16']),
       heads=st.integers(1, 9), n_rows=st.sampled_from([None, 1, 5, 12, 40, 77]), pick=st.sampled_from(['all', 'head', 'layer', 'factor']),
       seed=st.integers(0, 2 ** 16))
def test_finalize_property(sides, acc, heads, n_rows, pick, seed):
    """Any mix of map sizes, any dtype of the sums (fp16 / bf16 / f32: each has its own form of the matrix-core x2 kernel), any row count
    (ABI v6) and any of the reference's selections (trace.py:113): mean over keys of clamp(bicubic(plane)) matches the oracle; linear in a
    positive power-of-two scale of the inputs (clamp and bicubic are positively homogeneous)."""
    rng = np.random.default_rng(seed)
    accumulate = 'float32' if acc == 'float32' else 'exact'
    np_dt = ho.BF16 if acc == 'bfloat16' else acc
    eng = _engine(n_layers=len(sides), accumulate=accumulate)
    eng2 = _engine(n_layers=len(sides), accumulate=accumulate)
    raw = []
    for layer, side in enumerate(sides):
        planes = rng.standard_normal((2 * heads, side * side, 77)).astype(np.float32) * 2
        planes = ho.round_bf16(planes) if acc == 'bfloat16' else planes.astype(acc)
        factor = 64 // side if side <= 64 else 0
        eng.tap_probs(layer, _dev(planes, np_dt), factor=factor)
        eng2.tap_probs(layer, _dev(planes * 2, np_dt), factor=factor)
        raw += [((factor, layer, h), ho.unravel(planes)[h]) for h in range(heads)]
    kw = {'all': {}, 'head': dict(head_idx=heads - 1), 'layer': dict(layer_idx=len(sides) - 1), 'factor': dict(factors=[raw[0][0][0]])}[pick]
    want = ho.global_heat_map(raw, 4096, **kw)[:n_rows]
    got = eng.global_heat_map(n_rows=n_rows, **kw).cpu().numpy()
    got2 = eng2.global_heat_map(n_rows=n_rows, **kw).cpu().numpy()
    eng.close()
    eng2.close()
    assert got.shape == want.shape
    tol = 3e-6 * max(1.0, np.abs(want).max())
    np.testing.assert_allclose(got, want, rtol=0, atol=tol)
    np.testing.assert_allclose(got2, 2 * got, rtol=0, atol=4 * tol)


def test_trace_prompts_single_process():
    """daam_amd.distributed.trace_prompts without a process group = plain loop over the prompts."""
    from daam_amd.distributed import trace_prompts
    z, meta = load_golden('sd15_f16')
    pipe = golden_pipe(meta, device=DEV)
    prompts = ['a dog', 'a photo of a monkey']
    maps, rows = trace_prompts(pipe, prompts, num_inference_steps=3)
    assert maps.shape == (2, 77, 64, 64) and rows == [4, 7]
    import daam_amd
    with daam_amd.trace(pipe) as tc:
        pipe(prompts[1], num_inference_steps=3)
        ref = tc.compute_global_heat_map().heat_maps
    assert torch.allclose(maps[1, :rows[1]], ref, rtol=0, atol=1e-6)


def test_to_experiment_and_plot(tmp_path):
    """trace.to_experiment (reference trace.py:68-81) -> save / load in the reference's layout, and the
    overlay plot of a word map on a real image."""
    import PIL.Image
    import daam_amd
    z, meta = load_golden('sd15_f16')
    pipe = golden_pipe(meta, device=DEV)
    with daam_amd.trace(pipe) as tc:
        pipe('a dog', num_inference_steps=2)
        tc.last_image = PIL.Image.fromarray(np.full((64, 64, 3), 128, dtype=np.uint8))
        exp = tc.to_experiment(str(tmp_path), seed=3, id='g0', subtype='run')
        ghm = tc.compute_global_heat_map()
    assert exp.global_heat_map.shape == (4, 64, 64) and exp.prompt == 'a dog'
    exp.save()
    root = tmp_path / 'g0'
    assert (root / 'run' / 'generation.pt').exists() and (root / 'run' / 'output.png').exists()
    assert (root / 'run' / 'dog.heat_map.png').exists()                   # save_all_heat_maps
    back = daam_amd.GenerationExperiment.load(root, subtype='run')
    # two finalize calls: the f32 atomics may add the keys in a different order
    assert torch.allclose(back.global_heat_map, ghm.heat_maps.cpu(), rtol=0, atol=1e-6)
    # a LOADED experiment (map on the CPU) still yields word maps and overlay files, like the reference's
    assert back.global_heat_map.device.type == 'cpu'
    whm = back.heat_map().compute_word_heat_map('dog')
    assert torch.allclose(whm.heatmap.cpu(), ghm.compute_word_heat_map('dog').heatmap.cpu(), rtol=0, atol=1e-6)
    (root / 'run' / 'dog.heat_map.png').unlink()
    saved = back.save_all_heat_maps()
    assert set(saved) == {'a', 'dog'} and all(p.exists() for p in saved.values())


"""Full-size layers of BASELINE.json's configurations through 50 denoising steps (one deferred launch with the fp16
running sums carried in registers), compared NUMERICALLY with the numpy oracle (``oracle/heatmap_oracle.py``, pinned to
the reference's golden vectors) -- raw running sums element by element and the finalized global map.  The oracle
takes a few seconds per case at these sizes.  Run with ``-m gpu`` on an MI355X.

Tolerances (fp16 pipeline, fp16 sums = the reference's arithmetic, heatmap.py:156).  The HIP path and the oracle round
at the same points (fp16 logits, f32 softmax, fp16 probabilities, fp16 add); they differ in the f32 summation order of
q.k, which moves a logit across an fp16 rounding boundary now and then (0.02 - 0.06 % of the elements).  One ulp of a logit
of magnitude 8 .. 32 is 2^-7 .. 2^-6, so that step's probability -- and, when the flipped logit is the row's dominant one,
every probability of the row -- changes by up to e^(2^-6) - 1 = 1.6 % (measured with tools/exp/debug_ulps.py: the compensated
and the fast softmax, immediate and deferred launches all show the same elements).  Any two correct implementations of
the reference's arithmetic (rocBLAS vs MKL GEMM order) differ like this.  Hence:
  * running sums, every element: |diff| <= 2^-6 |value| + 2 ulp(value);  whole key: within 1 ulp of its largest sum;
    at most 1 % of the elements differ at all;
  * global map (bicubic -> clamp -> mean over the layer's keys): <= 1e-3 max-abs (BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import heatmap_oracle as ho

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _ulp16(x):
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -14)))
    return 2.0 ** (e - 10)


def _to_bh(x, heads):
    b, s, c = x.shape
    d = c // heads
    return np.ascontiguousarray(x.reshape(b, s, heads, d).transpose(0, 2, 1, 3)).reshape(b * heads, s, d)


CASES = [
    # name, heads, side, head_dim, steps, latent_hw (-> factor), n_q (distinct query sets, cycled), steps per launch
    ('sdxl1024_64x64_H10', 10, 64, 64, 50, 4096, 50, 64),     # SDXL-1024 up_blocks[1] / down_blocks[1]
    ('sdxl1024_32x32_H20', 20, 32, 64, 50, 4096, 50, 64),     # SDXL-1024 up_blocks[0] / down_blocks[2]
    ('sdxl2048_128x128_H4', 4, 128, 64, 20, 4096, 20, 64),    # SDXL-2048: hw = 16384, factor 0 (bicubic x0.5); 4 of the 10 heads
    # SDXL-2048 as BASELINE.json configs[4] runs it: all 10 heads, 100 steps, launches of 24 steps (what a 32 GiB byte
    # budget gives): every launch after the first reads the fp16 sums back (fresh = 0) -- 5 launches
    ('sdxl2048_128x128_H10_100steps_5launches', 10, 128, 64, 100, 4096, 25, 24),
    ('sd15_16x16_d160', 8, 16, 160, 50, 4096, 50, 64),        # SD-v1.5 deepest level
    ('sd15_32x32_d80', 8, 32, 80, 50, 4096, 50, 64),
    ('sd15_64x64_d40', 8, 64, 40, 50, 4096, 50, 64),          # SD-v1.5 outer level: head_dim 40 zero-padded to 64 (FULL64 = false)
]


@pytest.mark.parametrize('name,heads,side,d,steps,latent_hw,n_q,defer', CASES, ids=[c[0] for c in CASES])
def test_full_size_layer_50_steps_vs_oracle(name, heads, side, d, steps, latent_hw, n_q, defer):
    from daam_amd.engine import HeatMapEngine
    hw = side * side
    rng = np.random.default_rng(hw + d)
    scale = d ** -0.5
    # SOS-dominant keys: token 0's sum approaches the step count, where 1 fp16 ulp = 2^-5 .. 2^-6
    k = rng.standard_normal((2, 77, heads * d)).astype(np.float32)
    k[:, 0] *= 3.0
    k = k.astype(np.float16)
    qs = [rng.standard_normal((2, hw, heads * d)).astype(np.float16) for _ in range(n_q)]
    # a persistent component along token 0's key so that its probability stays high step after step
    k0 = k[1, 0].astype(np.float32).reshape(heads, d)
    for q in qs:
        qq = q.reshape(2, hw, heads, d)
        qq[1] += (0.35 * k0 / np.sqrt((k0 ** 2).mean(-1, keepdims=True)))[None].astype(np.float16)

    factor = ho.layer_factor(latent_hw, hw)
    raw = ho.RawMaps(np.float16)
    k_bh = _to_bh(k, heads)
    # the probabilities of a query set are computed once (get_attention_scores, trace.py:276; only the conditional half
    # is kept by _unravel_attn, trace.py:240: the unconditional half is left zero here) and added once per step in which
    # the set recurs -- the reference's step-by-step fp16 adds (heatmap.py:156) in the same order
    probs = []
    for q in qs:
        p_cond = ho.attention_probs(_to_bh(q, heads)[heads:], k_bh[heads:], scale, np.float16)
        probs.append(np.concatenate([np.zeros_like(p_cond), p_cond]))
    for s in range(steps):
        ho.tap(raw, 0, None, None, scale, latent_hw=latent_hw, pipe_dtype=np.float16, probs=probs[s % n_q])
    del probs
    want = np.stack([v for _, v in raw]).astype(np.float64)        # [heads, 77, side, side]
    assert want.shape == (heads, 77, side, side) and want.max() > 0.5 * steps

    eng = HeatMapEngine(1, tokens=77, out_side=int(np.sqrt(latent_hw)), accumulate='exact', defer_steps=defer)
    kd = torch.from_numpy(k).to(DEV)
    qd = [torch.from_numpy(q).to(DEV) for q in qs]
    for s in range(steps):
        eng.tap_qk(0, qd[s % n_q], kd, heads, scale, factor=factor)
    if steps <= defer:
        assert eng.pending_taps == steps                           # one launch for the whole generation
    else:
        assert eng.pending_taps == steps % defer or eng.pending_taps == defer
    items = list(eng.items())
    assert eng.last_flush()['launches'] == -(-steps // defer)      # later launches read the sums back (TapLayer.fresh = 0)
    assert [kk for kk, _ in items] == [(factor, 0, h) for h in range(heads)]
    got = torch.stack([v for _, v in items])
    assert got.dtype == torch.float16
    got = got.float().cpu().numpy().astype(np.float64)

    diff = np.abs(got - want)
    excess = diff - (2.0 ** -6 * np.abs(want) + 2 * _ulp16(np.maximum(np.abs(got), np.abs(want))))
    assert excess.max() <= 0, f'{name}: running sums off by {diff.flat[np.argmax(excess)]} at value {want.flat[np.argmax(excess)]}'
    # a whole key: within 1 ulp of its largest sum; 2 when query sets recur (a probability that differs by one fp16 ulp then
    # differs in every recurrence, and the two fp16 accumulation chains can round apart at more than one of their steps)
    key_ulps = 1 if n_q >= steps else 2
    for h in range(heads):
        assert diff[h].max() <= key_ulps * _ulp16(np.asarray(want[h].max())) + 1e-12, f'{name}: head {h}'
    assert (diff > 0).mean() <= 0.01
    np.testing.assert_allclose(got.sum(1), steps, atol=steps * 77 * 2.0 ** -11)

    gm = eng.global_heat_map().cpu().numpy()
    ref = ho.global_heat_map(list(raw), latent_hw)
    assert gm.shape == ref.shape
    # ONE layer = `heads` keys: a key whose sum differs by one fp16 ulp (2^-5 for sums in [32, 64)) moves this mean by
    # ulp / heads; the north_star's 1e-3 is a statement about the mean over all 1100 keys of a generation (there the same
    # key moves it by 3e-5: tests/test_gpu_integration.py measures 1e-5 on the full stack)
    tol = max(1e-3, 1.01 * float(_ulp16(np.asarray(want.max()))) / heads)          # + fp32 rounding of the mean itself
    assert np.abs(gm - ref).max() <= tol, f'{name}: global map {np.abs(gm - ref).max()} > {tol}'
    eng.close()


def test_sdxl_50_step_generation_two_resolutions_one_launch():
    """Both SDXL layer classes in ONE 50-step launch (the bench configuration's launch structure, fewer layers):
    maps of the two resolutions against the oracle through finalize (same-size class + x2 class side by side)."""
    from daam_amd.engine import HeatMapEngine
    rng = np.random.default_rng(99)
    steps, d = 50, 64
    layers = [(0, 20, 32), (1, 10, 64), (2, 20, 32)]
    raw = ho.RawMaps(np.float16)
    eng = HeatMapEngine(len(layers), tokens=77, out_side=64, accumulate='exact', defer_steps=64)
    data = {}
    for li, heads, side in layers:
        k = rng.standard_normal((2, 77, heads * d)).astype(np.float32)
        k[:, 0] *= 3.0
        k = k.astype(np.float16)
        qs = [rng.standard_normal((2, side * side, heads * d)).astype(np.float16) for _ in range(3)]
        data[li] = (heads, side, k, qs, torch.from_numpy(k).to(DEV), [torch.from_numpy(q).to(DEV) for q in qs])
    for s in range(steps):
        for li, heads, side in layers:
            _, _, k, qs, kd, qd = data[li]
            ho.tap(raw, li, _to_bh(qs[s % 3], heads), _to_bh(k, heads), d ** -0.5, latent_hw=4096, pipe_dtype=np.float16)
            eng.tap_qk(li, qd[s % 3], kd, heads, d ** -0.5, factor=64 // side)
    assert eng.pending_taps == steps * len(layers)
    top = max(float(v.max()) for _, v in raw)
    for kw, n_sel in ((dict(), 50), (dict(factors=[2]), 40), (dict(layer_idx=1), 10), (dict(head_idx=7), 3)):
        gm = eng.global_heat_map(**kw).cpu().numpy()
        ref = ho.global_heat_map(list(raw), 4096, **kw)
        # all keys: the north_star bound; a selection of n keys averages fewer of the (rare) 1-ulp-of-the-sum differences:
        # one such key moves the mean by ulp(largest sum) / n  (this test cycles 3 query sets, so a flipped rounding recurs)
        tol = 1e-3 if not kw else max(1e-3, float(_ulp16(np.asarray(top))) / n_sel)
        assert np.abs(gm - ref).max() <= tol, (kw, np.abs(gm - ref).max(), tol)
    eng.close()

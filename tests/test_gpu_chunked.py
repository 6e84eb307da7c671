"""The chunked tap kernel (daam_amd/csrc/daam_tap_chunk.hip -- any head_dim in 64-element chunks, one
kind of workgroup per flush) against the kernels it stands in for (tap_d64_kernel, tap_wide_kernel<3|5>): same tiling, same
k order, same softmax, so the running sums must be BIT-IDENTICAL -- per layer shape, immediate and deferred, across launches,
and for a whole SD-v1.5-shaped flush (head_dim 40 / 80 / 160 in ONE launch instead of three kernels side by side: the default
for launches that mix head dims; ``DAAM_TAP_CHUNKED=1`` puts every fp16 layer on it, ``=0`` none).
The default kernels are what the rest of the suite pins to the oracle and the reference's goldens.
Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _engine(monkeypatch, chunked, n_layers, accumulate, defer):
    """``chunked``: '0' = never (the specialised kernels), '1' = every fp16 layer, None = the default (launches that mix head dims)."""
    from daam_amd import engine as E
    E.release_parked_contexts()                       # the switch is read when a native context is created
    monkeypatch.setenv('DAAM_TAP_SLAB', '0')           # this file is about the chunked kernel; the slab kernel has tests/test_gpu_slab.py
    if chunked is None:
        monkeypatch.delenv('DAAM_TAP_CHUNKED', raising=False)
    else:
        monkeypatch.setenv('DAAM_TAP_CHUNKED', chunked)
    return E.HeatMapEngine(n_layers, tokens=77, out_side=64, accumulate=accumulate, defer_steps=defer)


def _inputs(shapes, steps, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    sets = []
    for _ in range(steps):
        cur = []
        for (heads, hw, d) in shapes:
            q = torch.randn(2, hw, heads * d, generator=g, device=DEV, dtype=torch.float16)
            k = torch.randn(2, 77, heads * d, generator=g, device=DEV, dtype=torch.float16)
            k[:, 0, :] *= 2.0
            cur.append((q, k))
        sets.append(cur)
    return sets


def _run(eng, shapes, sets, rounds=1):
    """``rounds`` x (all steps, then a flush): the second round reads the sums the first one wrote."""
    for _ in range(rounds):
        for cur in sets:
            for layer, ((heads, hw, d), (q, k)) in enumerate(zip(shapes, cur)):
                side = int(round(hw ** 0.5))                        # the engine keeps [heads, 77, side, side] sums
                factor = max(1, 64 // side)
                eng.tap_qk(layer, q, k, heads, d ** -0.5, factor)
        eng.flush()
    torch.cuda.synchronize()
    return {key: t.clone() for key, t in eng.items()}, eng.last_flush()


LAYER_CASES = [
    # heads, hw, head_dim
    (8, 4096, 40),       # SD-v1.5 64 x 64: one partial chunk (5 of 8 pieces: k-step 1 masked)
    (8, 1024, 80),       # SD-v1.5 32 x 32: chunk 1 holds 2 pieces (k-step 0 masked, k-step 1 skipped)
    (8, 256, 160),       # SD-v1.5 16 x 16: chunk 2 holds 4 pieces (k-step 1 skipped, no mask)
    (10, 1024, 64),      # SDXL: one full chunk
    (2, 64, 8),          # one piece; tile larger than the layer (waves 2, 3 outside)
    (4, 256, 96),        # 64 + 32
    (2, 256, 128),       # two full chunks
    (2, 144, 48),        # 12 x 12: a full tile + 16 pixels (waves 1..3 of the second tile outside)
    (3, 576, 120),       # 24 x 24: 64 + 56 (7 pieces), 4.5 tiles
]


@pytest.mark.parametrize('accumulate', ['exact', 'float32'])
@pytest.mark.parametrize('defer', [0, 8])
def test_chunked_layers_bit_identical_to_default_kernels(monkeypatch, accumulate, defer):
    steps = 3
    sets = _inputs(LAYER_CASES, steps, seed=5)
    ref_eng = _engine(monkeypatch, '0', len(LAYER_CASES), accumulate, defer)
    ref, _ = _run(ref_eng, LAYER_CASES, sets, rounds=2)
    ref_eng.close()
    got_eng = _engine(monkeypatch, '1', len(LAYER_CASES), accumulate, defer)
    got, flush = _run(got_eng, LAYER_CASES, sets, rounds=2)
    got_eng.close()
    assert set(got) == set(ref)
    if defer:
        assert flush['kernels'] == 1 and flush['side_streams'] == 0, flush      # every head_dim in one launch
    for key in ref:
        assert float(ref[key].float().abs().sum()) > 0, key
        assert torch.equal(got[key], ref[key]), (key, float((got[key].float() - ref[key].float()).abs().max()))


def test_chunked_sd15_flush_is_one_kernel_and_bit_identical(monkeypatch):
    """The SD-v1.5 layer set (15 layers, head_dim 40 / 80 / 160, execution order) x 20 steps in one deferred launch."""
    s = [16, 32, 64]
    up = [(8, s[i // 3] ** 2, [160, 80, 40][i // 3]) for i in range(9)]
    down = [(8, [64, 32, 16][i // 2] ** 2, [40, 80, 160][i // 2]) for i in range(6)]
    shapes = down + up
    sets = _inputs(shapes, 20, seed=9)
    ref_eng = _engine(monkeypatch, '0', len(shapes), 'exact', 64)
    ref, rflush = _run(ref_eng, shapes, sets)
    ref_eng.close()
    assert rflush['kernels'] == 3, rflush
    got_eng = _engine(monkeypatch, None, len(shapes), 'exact', 64)     # the default: this launch mixes head dims
    got, flush = _run(got_eng, shapes, sets)
    got_eng.close()
    assert flush['kernels'] == 1 and flush['side_streams'] == 0 and flush['max_steps'] == 20, flush
    assert len(got) == 120
    for key in ref:
        assert torch.equal(got[key], ref[key]), key
    # the sums of a pixel over the 77 tokens: every step's probabilities add up to 1
    tot = np.stack([got[key].float().sum(0).cpu().numpy().ravel()[:64] for key in list(got)[:8]])
    assert np.abs(tot - 20).max() < 0.25


@pytest.mark.parametrize('accumulate', ['exact', 'float32'])
def test_chunked_bf16_layers(monkeypatch, accumulate):
    """bf16 pipelines: head_dim <= 64 bit-identical to the bf16 head_dim-64 kernel; wider heads (DAAM_TAP_CHUNKED=0 leaves them on the
    any-shape kernel: f32 FMA dot products, another summation order) within two bf16 ulps of the sums.  First run on the chip in
    round 4; since then the DEFAULT for bf16 layers with head_dim > 64, and a bf16 launch that mixes head dims is one kernel."""
    shapes = [(8, 1024, 64), (8, 4096, 40), (8, 1024, 80), (8, 256, 160), (2, 144, 48)]
    g = torch.Generator(device=DEV).manual_seed(3)
    sets = [[(torch.randn(2, hw, h * d, generator=g, device=DEV).bfloat16(), torch.randn(2, 77, h * d, generator=g, device=DEV).bfloat16())
             for (h, hw, d) in shapes] for _ in range(3)]
    ref_eng = _engine(monkeypatch, '0', len(shapes), accumulate, 8)
    ref, _ = _run(ref_eng, shapes, sets, rounds=2)
    ref_eng.close()
    got_eng = _engine(monkeypatch, '1', len(shapes), accumulate, 8)
    got, flush = _run(got_eng, shapes, sets, rounds=2)
    got_eng.close()
    assert flush['kernels'] == 1, flush
    for key in ref:
        d = shapes[key[1]][2]
        a, b = got[key].float(), ref[key].float()
        if d <= 64:
            assert torch.equal(got[key], ref[key]), (key, float((a - b).abs().max()))
        else:
            tol = 2.0 ** -7 * b.abs() + 2.0 ** -9 if accumulate == 'exact' else 2.0 ** -8 * b.abs() + 2.0 ** -9
            assert bool(((a - b).abs() <= tol).all()), (key, float((a - b).abs().max()))
    dflt_eng = _engine(monkeypatch, None, len(shapes), accumulate, 8)
    dflt, dflush = _run(dflt_eng, shapes, sets, rounds=2)
    dflt_eng.close()
    assert dflush['kernels'] == 1 and dflush['side_streams'] == 0, dflush
    for key in got:
        assert torch.equal(got[key], dflt[key]), key

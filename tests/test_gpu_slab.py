"""The slab tap kernel (daam_amd/csrc/daam_tap_slab.hip -- deferred fp16 layers of head_dim 40 / 80 / 160, a 640-byte slab of adjacent
heads per workgroup, whole 128-byte lines of Q) against the kernels it stands in for: the chunked kernel (``DAAM_TAP_SLAB=0``) and the
specialised ones (``DAAM_TAP_CHUNKED=0``).  Same operand layout, k order, MFMA chain and softmax: the running sums must be
BIT-IDENTICAL -- per layer shape, across launches (the second launch reads the sums the first one wrote), for fp16 and f32 sums,
and for a whole SD-v1.5-shaped launch; shapes the slab kernel does not take fall through to the other kernels unchanged.
The default kernels are what the rest of the suite pins to the oracle and the reference's goldens.  Run with ``-m gpu`` on an MI355X."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = 'cuda:0'


def _engine(monkeypatch, slab, n_layers, accumulate, defer, chunked=None, tail=None):
    from daam_amd import engine as E
    E.release_parked_contexts()                       # the switches are read when a native context is created
    monkeypatch.setenv('DAAM_TAP_SLAB', '1' if slab else '0')
    if tail is None:
        monkeypatch.delenv('DAAM_SLAB_TAIL', raising=False)
    else:
        monkeypatch.setenv('DAAM_SLAB_TAIL', str(tail))
    if chunked is None:
        monkeypatch.delenv('DAAM_TAP_CHUNKED', raising=False)
    else:
        monkeypatch.setenv('DAAM_TAP_CHUNKED', chunked)
    return E.HeatMapEngine(n_layers, tokens=77, out_side=64, accumulate=accumulate, defer_steps=defer)


def _inputs(shapes, steps, seed, batch=2):
    g = torch.Generator(device=DEV).manual_seed(seed)
    sets = []
    for _ in range(steps):
        cur = []
        for (heads, hw, d) in shapes:
            q = torch.randn(batch, hw, heads * d, generator=g, device=DEV, dtype=torch.float16)
            k = torch.randn(batch, 77, heads * d, generator=g, device=DEV, dtype=torch.float16)
            k[:, 0, :] *= 2.0
            cur.append((q, k))
        sets.append(cur)
    return sets


def _run(eng, shapes, sets, rounds=1):
    for _ in range(rounds):
        for cur in sets:
            for layer, ((heads, hw, d), (q, k)) in enumerate(zip(shapes, cur)):
                side = int(round(hw ** 0.5))
                eng.tap_qk(layer, q, k, heads, d ** -0.5, max(1, 64 // side))
        eng.flush()
    torch.cuda.synchronize()
    return {key: t.clone() for key, t in eng.items()}, eng.last_flush(), _last_launch(eng)


def _last_launch(eng):
    from daam_amd import _native as nat
    grid, block, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    nat.check(eng.lib.daam_last_launch(eng.ctx, 0, ctypes.byref(grid), ctypes.byref(block), ctypes.byref(lds)))
    return dict(grid=grid.value, block=block.value, lds=lds.value)


SLAB_CASES = [
    # heads, hw, head_dim
    (8, 4096, 40),       # SD-v1.5 64 x 64: one slab of eight heads, 128 tiles
    (8, 1024, 80),       # 32 x 32: two slabs of four heads
    (8, 256, 160),       # 16 x 16: four slabs of two heads, waves 4..7 fetch only
    (8, 64, 160),        # the mid block: two tiles
    (16, 256, 40),       # two slabs of eight heads
    (8, 144, 80),        # 12 x 12 (768-px SD-v1.x): 4.5 tiles -- the last tile's second half lies outside the layer
    (8, 16, 40),         # 4 x 4: half a tile
    (4, 2304, 160),      # 48 x 48, two slabs
    (8, 144, 40),        # hw % 32 == 16 with a tail: the main entries end / the tail entry begins on a 16-pixel (not a 32-pixel) boundary
    (8, 400, 40),        # 20 x 20: the same, 12.5 tiles
]


@pytest.mark.parametrize('tail', [None, 0, 100, 60])
@pytest.mark.parametrize('accumulate', ['exact', 'float32'])
def test_slab_layers_bit_identical_to_chunked_kernel(monkeypatch, accumulate, tail):
    """``tail`` = DAAM_SLAB_TAIL: the share of a head_dim-40 layer's pixels taken by half-size (16-pixel) workgroups at the end of the
    launch (default 25 %; 0 = none, 100 = all of them): the same sums bit for bit."""
    steps = 5
    sets = _inputs(SLAB_CASES, steps, seed=11)
    ref_eng = _engine(monkeypatch, False, len(SLAB_CASES), accumulate, 8)
    ref, rflush, rlaunch = _run(ref_eng, SLAB_CASES, sets, rounds=2)
    ref_eng.close()
    assert rlaunch['block'] == 256, rlaunch                          # the chunked kernel
    got_eng = _engine(monkeypatch, True, len(SLAB_CASES), accumulate, 8, tail=tail)
    got, flush, launch = _run(got_eng, SLAB_CASES, sets, rounds=2)
    got_eng.close()
    assert flush['kernels'] == 1 and flush['side_streams'] == 0 and flush['max_steps'] == steps, flush
    assert launch['block'] == 512 and launch['lds'] >= 71 * 1024, launch   # the slab kernel took the launch
    assert set(got) == set(ref)
    for key in ref:
        assert float(ref[key].float().abs().sum()) > 0, key
        assert torch.equal(got[key], ref[key]), (key, float((got[key].float() - ref[key].float()).abs().max()))


def test_slab_sd15_launch_one_kernel_bit_identical_to_specialised_kernels(monkeypatch):
    """The SD-v1.5 layer set (16 layers incl. the mid block, execution order) x 50 steps in one deferred launch: slab kernel against
    the three specialised kernels side by side (DAAM_TAP_CHUNKED=0)."""
    s = [16, 32, 64]
    up = [(8, s[i // 3] ** 2, [160, 80, 40][i // 3]) for i in range(9)]
    down = [(8, [64, 32, 16][i // 2] ** 2, [40, 80, 160][i // 2]) for i in range(6)]
    shapes = down + [(8, 64, 160)] + up
    sets = _inputs(shapes, 50, seed=9)
    ref_eng = _engine(monkeypatch, False, len(shapes), 'exact', 64, chunked='0')
    ref, rflush, _ = _run(ref_eng, shapes, sets)
    ref_eng.close()
    assert rflush['kernels'] == 3, rflush
    got_eng = _engine(monkeypatch, True, len(shapes), 'exact', 64)
    got, flush, launch = _run(got_eng, shapes, sets)
    names = got_eng.last_kernels(0)
    got_eng.close()
    assert flush['kernels'] == 1 and flush['side_streams'] == 0 and flush['max_steps'] == 50, flush
    assert launch['block'] == 512 and names == 'tap_slab_kernel', (launch, names)
    # segments: 168 head_dim-160 workgroups, 5 x 96 of head_dim 40 (32-pixel tiles: the first three quarters of each layer), 320 of head_dim
    # 80, 5 x 64 half-size ones (16-pixel tiles: the last quarter of the head_dim-40 layers); an eighth of each segment per XCD
    assert launch['grid'] == 8 * (21 + 60 + 40 + 40), launch
    assert len(got) == 128
    for key in ref:
        assert torch.equal(got[key], ref[key]), key
    tot = np.stack([got[key].float().sum(0).cpu().numpy().ravel()[:64] for key in list(got)[:8]])
    assert np.abs(tot - 50).max() < 0.5                               # every step's probabilities add up to 1


def test_slab_falls_through_for_shapes_it_does_not_take(monkeypatch):
    """No CFG (batch 1: the kept heads start in the middle of a slab), other head dims beside supported ones: such layers stay on the
    other kernels -- in the same flush as slab layers -- and every sum matches the chunked-only run."""
    shapes = [(8, 1024, 80), (8, 256, 64), (6, 576, 48)]
    sets = _inputs(shapes, 4, seed=4)
    ref_eng = _engine(monkeypatch, False, len(shapes), 'exact', 8)
    ref, _, _ = _run(ref_eng, shapes, sets, rounds=2)
    ref_eng.close()
    got_eng = _engine(monkeypatch, True, len(shapes), 'exact', 8)
    got, flush, _ = _run(got_eng, shapes, sets, rounds=2)
    got_eng.close()
    assert flush['kernels'] >= 2, flush                              # slab layers + the rest
    for key in ref:
        assert torch.equal(got[key], ref[key]), key
    # batch 1: BH / 2 = 4 heads kept of 8 -- not a whole slab of eight: the chunked kernel takes the layer
    shapes1 = [(8, 1024, 40)]
    sets1 = _inputs(shapes1, 3, seed=6, batch=1)
    a_eng = _engine(monkeypatch, False, 1, 'exact', 8)
    a, _, _ = _run(a_eng, shapes1, sets1)
    a_eng.close()
    b_eng = _engine(monkeypatch, True, 1, 'exact', 8)
    b, _, blaunch = _run(b_eng, shapes1, sets1)
    b_eng.close()
    assert blaunch['block'] == 256, blaunch
    for key in a:
        assert torch.equal(a[key], b[key]), key

"""``daam_attend`` (include/daam_hip.h): the processor's attention -- ``get_attention_scores`` + ``bmm`` +
``batch_to_head_dim`` of the reference's ``UNetCrossAttentionHooker.__call__`` (daam/trace.py:276,296-297) -- with the
heat-map tap (:289-294, heatmap.py:153-156) fused into the same kernel.  Checked against

* the numpy oracle (``oracle/heatmap_oracle.py::attention_output``: logits -> fp16, softmax, probabilities -> fp16, value product
  accumulated wide -> fp16; pinned on the CPU to what the unmodified reference's processor returned),
* the reference's own sequence of torch ops run in PyTorch-ROCm eager on the same inputs (``oracle/torch_hooks.py``),
* the library's stand-alone tap (``daam_tap_qk``): the fused tap must leave BIT-IDENTICAL running sums.

Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest
import torch

from oracle import torch_hooks as th

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _inputs(batch, heads, hw, seed, gain=1.0, dtype=torch.float16, head_dim=64):
    g = torch.Generator(device='cpu').manual_seed(seed)
    c = heads * head_dim
    q = (torch.randn(batch, hw, c, generator=g) * gain).to(dtype).to(DEV)
    k = torch.randn(batch, 77, c, generator=g)
    k[:, 0] *= 3.0                                         # start-of-text dominance, like real cross-attention
    k = (k * gain).to(dtype).to(DEV)
    v = torch.randn(batch, 77, c, generator=g).to(dtype).to(DEV)
    return q, k, v


def _to_heads(t, heads):                                   # diffusers head_to_batch_dim
    b, s, c = t.shape
    return t.reshape(b, s, heads, c // heads).permute(0, 2, 1, 3).reshape(b * heads, s, c // heads)


def _from_heads(t, heads):                                 # diffusers batch_to_head_dim
    bh, s, d = t.shape
    return t.reshape(bh // heads, heads, s, d).permute(0, 2, 1, 3).reshape(bh // heads, s, d * heads)


def _reference_eager(q, k, v, heads, scale):
    """The reference's processor between the projections: trace.py:272-276, 296-297 (torch ops, on the GPU)."""
    qh, kh, vh = (_to_heads(t, heads) for t in (q, k, v))
    probs = th.attention_probs(qh, kh, scale)
    return _from_heads(torch.bmm(probs, vh), heads), probs


def _restated_f64(q, k, v, heads, scale, round_logits=True):
    """The numpy oracle (``oracle/heatmap_oracle.py::attention_output``, pinned to the processor outputs of the unmodified
    reference by ``tests/test_oracle_golden.py``): the reference's rounding points, wide accumulation."""
    from oracle import heatmap_oracle as ho
    qh, kh, vh = (_to_heads(t, heads).cpu().numpy() for t in (q, k, v))
    probs = ho.attention_probs(qh, kh, scale, np.float16, upcast_attention=not round_logits)
    out = ho.attention_output(qh, kh, vh, scale, np.float16, upcast_attention=not round_logits)
    return torch.from_numpy(ho.batch_to_head_dim(out, heads)), probs


def _engine(n_layers=1, accumulate='exact', defer_steps=0):
    from daam_amd.engine import HeatMapEngine
    return HeatMapEngine(n_layers, defer_steps=defer_steps, accumulate=accumulate)


@pytest.mark.parametrize('batch,heads,hw', [(2, 10, 4096), (2, 20, 1024), (2, 3, 576), (1, 4, 64), (2, 2, 16384)])
def test_attend_output_matches_reference(batch, heads, hw):
    q, k, v = _inputs(batch, heads, hw, seed=hw + heads)
    eng = _engine()
    out = eng.attend(0, q, k, v, heads, 0.125, 1, True, tapped=False)
    assert out is not None and out.shape == q.shape and out.dtype == torch.float16
    want_eager, _ = _reference_eager(q, k, v, heads, 0.125)
    scale_v = want_eager.float().abs().max().item()
    err_eager = (out.float() - want_eager.float()).abs().max().item()
    # fp16 rounding of the output (2^-11 relative) + an fp16-boundary flip of a logit now and then (summation order)
    assert err_eager <= 2e-3 * scale_v, f'vs the reference ops in eager: {err_eager / scale_v:.2e}'
    if hw <= 4096:
        want, _ = _restated_f64(q, k, v, heads, 0.125)
        err = (out.float().cpu() - want.float()).abs()
        assert err.max().item() <= 2e-3 * scale_v
        # ... and nearly everywhere the two agree to the last fp16 bit or one ulp
        ulp = np.spacing(np.abs(want.numpy()).astype(np.float16)).astype(np.float64)
        assert (err.numpy() <= ulp).mean() >= 0.995
    eng.close()


@pytest.mark.parametrize('accumulate', ['exact', 'float32'])
@pytest.mark.parametrize('batch,heads,hw', [(2, 10, 4096), (2, 5, 1024), (1, 4, 256), (2, 3, 576)])
def test_fused_tap_leaves_the_sums_of_the_stand_alone_tap(batch, heads, hw, accumulate):
    steps = 3
    fused, plain = _engine(accumulate=accumulate), _engine(accumulate=accumulate)
    for step in range(steps):
        q, k, v = _inputs(batch, heads, hw, seed=100 * step + heads)
        out = fused.attend(0, q, k, v, heads, 0.125, 1, True, tapped=True)
        assert out is not None
        plain.tap_qk(0, q, k, heads, 0.125, 1, True)
        want, _ = _reference_eager(q, k, v, heads, 0.125)
        assert (out.float() - want.float()).abs().max().item() <= 2e-3 * want.float().abs().max().item()
    a, b = dict(fused.items()), dict(plain.items())
    assert list(a) == list(b) and len(a) == (batch * heads - (batch * heads) // 2)
    for key in a:
        assert a[key].dtype == b[key].dtype
        assert torch.equal(a[key], b[key]), f'{key}: fused and stand-alone tap differ'
    # and the sums are the reference's: probabilities of the kept heads, summed over the steps in the sum dtype
    q, k, v = _inputs(batch, heads, hw, seed=heads)        # step 0 again
    _, probs = _reference_eager(q, k, v, heads, 0.125)
    one = _engine(accumulate=accumulate)
    one.attend(0, q, k, v, heads, 0.125, 1, True, tapped=True)
    side = int(hw ** 0.5)
    kept = probs[probs.shape[0] // 2:]                                       # trace.py:240
    got = torch.stack([t for _, t in one.items()]).float()                   # [kept heads, 77, side, side]
    want = kept.permute(0, 2, 1).reshape(kept.shape[0], 77, side, side).float()
    diff = (got - want).abs()
    assert diff.max().item() <= 2.0 ** -6 * want.max().item() + 1e-3        # a flipped fp16 logit: e^(2^-6) - 1 relative
    assert (diff > 0).float().mean().item() <= 0.02
    for e in (fused, plain, one):
        e.close()


@pytest.mark.parametrize('accumulate', ['exact', 'float32'])
@pytest.mark.parametrize('head_dim,hw', [(40, 4096), (80, 1024), (160, 256), (8, 64), (96, 576), (128, 256)])
def test_attend_other_head_dims(head_dim, hw, accumulate):
    """SD-v1.5's head dims (40 / 80 / 160, 8 heads) and the corners of the three kernel shapes: output against the float64
    restatement and the reference ops, fused tap against the stand-alone tap (bit-identical: daam_tap_d64 / daam_tap_wide and
    daam_attend share the 16x16x32 tiling and the softmax code)."""
    heads, scale = 8, head_dim ** -0.5
    fused, plain = _engine(accumulate=accumulate), _engine(accumulate=accumulate)
    for step in range(2):
        q, k, v = _inputs(2, heads, hw, seed=7 * step + head_dim, head_dim=head_dim)
        out = fused.attend(0, q, k, v, heads, scale, 1, True, tapped=True)
        assert out is not None and out.shape == q.shape
        plain.tap_qk(0, q, k, heads, scale, 1, True)
        want, _ = _restated_f64(q, k, v, heads, scale)
        want_eager, _ = _reference_eager(q, k, v, heads, scale)
        ref_scale = want.float().abs().max().item()
        assert (out.float().cpu() - want.float()).abs().max().item() <= 2e-3 * ref_scale
        assert (out.float() - want_eager.float()).abs().max().item() <= 2e-3 * ref_scale
    a, b = dict(fused.items()), dict(plain.items())
    assert list(a) == list(b)
    for key in a:
        assert torch.equal(a[key], b[key]), key                  # same tiling, same MFMA order, same softmax code
    fused.close()
    plain.close()


@pytest.mark.parametrize('accumulate', ['exact', 'float32'])
@pytest.mark.parametrize('head_dim,heads,hw', [(64, 10, 4096), (64, 20, 1024), (40, 8, 4096), (80, 8, 1024), (160, 8, 256), (64, 3, 576)])
def test_attend_bf16_pipeline(head_dim, heads, hw, accumulate):
    """bf16 pipelines (round 3): ``daam_attend`` on ``v_mfma_f32_16x16x32_bf16`` with the reference's bf16 rounding points
    (bf16 logits, f32 softmax, bf16 probabilities, f32-accumulated value product rounded once to bf16).  Output against the
    numpy oracle (``ho.attention_output(..., BF16)``) and the reference's torch ops in eager; the fused tap against the
    stand-alone bf16 tap (bit-identical for head_dim <= 64, where both run the same tiling and softmax code) and against the
    oracle's sums.  Tolerances: a bf16 logit of magnitude 8 .. 16 has an ulp of 2^-4, so a logit that the f32 summation order
    moves across a rounding boundary (a few dozen of the millions per call) changes its probability by up to e^(2^-4) - 1 =
    6.4 % -- and every probability of the row when it is the dominant one: outputs within one bf16 ulp for >= 99 % of the
    elements and within 2^-4 of the largest everywhere; sums within 2^-3 |v| + one bf16 ulp, >= 99 % bit-equal."""
    from oracle import heatmap_oracle as ho
    scale = head_dim ** -0.5
    fused, plain = _engine(accumulate=accumulate), _engine(accumulate=accumulate)
    raw = ho.RawMaps(ho.BF16 if accumulate == 'exact' else np.float32)
    for step in range(2):
        q, k, v = _inputs(2, heads, hw, seed=11 * step + head_dim + heads, dtype=torch.bfloat16, head_dim=head_dim)
        out = fused.attend(0, q, k, v, heads, scale, 1, True, tapped=True)
        assert out is not None and out.shape == q.shape and out.dtype == torch.bfloat16
        plain.tap_qk(0, q, k, heads, scale, 1, True)
        qh, kh, vh = (_to_heads(t, heads).float().cpu().numpy() for t in (q, k, v))
        want = torch.from_numpy(ho.batch_to_head_dim(ho.attention_output(qh, kh, vh, scale, ho.BF16), heads))
        ref_scale = want.abs().max().item()
        err = (out.float().cpu() - want).abs()
        assert err.max().item() <= 2.0 ** -4 * ref_scale, err.max().item() / ref_scale
        ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.abs(want.numpy()), 1e-30))) - 7)        # bf16: 8 significant bits
        assert (err.numpy() <= ulp).mean() >= 0.99
        want_eager, _ = _reference_eager(q, k, v, heads, scale)
        err_eager = (out.float() - want_eager.float()).abs()
        assert err_eager.max().item() <= 2.0 ** -4 * ref_scale
        assert (err_eager <= 2.0 ** -7 * want_eager.float().abs().clamp_min(2.0 ** -20)).float().mean().item() >= 0.98
        ho.tap(raw, 0, qh, kh, scale, latent_hw=hw, pipe_dtype=ho.BF16)
    a, b = dict(fused.items()), dict(plain.items())
    assert list(a) == list(b) and len(a) == heads
    want_sums = np.stack([m for _, m in raw])
    got = torch.stack([t for _, t in fused.items()]).float().cpu().numpy()
    diff = np.abs(got - want_sums)
    assert (diff - (2.0 ** -3 * np.abs(want_sums) + 2.0 ** -7 * np.maximum(1.0, np.abs(want_sums)))).max() <= 0
    assert (diff > 0).mean() <= 0.01
    for key in a:
        assert a[key].dtype == b[key].dtype == (torch.bfloat16 if accumulate == 'exact' else torch.float32)
        if head_dim <= 64:
            assert torch.equal(a[key], b[key]), key
        else:                                                    # head_dim > 64: the stand-alone bf16 tap is the any-shape kernel
            d2 = (a[key].float() - b[key].float()).abs()
            assert (d2 - (2.0 ** -3 * b[key].float().abs() + 2.0 ** -7)).max().item() <= 0 and (d2 > 0).float().mean().item() <= 0.01
    fused.close()
    plain.close()


def test_attend_bf16_declines_unrounded_logits():
    q, k, v = _inputs(2, 4, 256, seed=3, dtype=torch.bfloat16)
    eng = _engine()
    assert eng.attend(0, q, k, v, 4, 0.125, 1, False, tapped=False) is None      # upcast_attention: the framework's attention
    assert eng.attend(0, q, k, v, 4, 0.125, 1, True, tapped=False) is not None
    eng.close()


def test_attend_unrounded_logits_and_general_scale():
    """``upcast_attention`` (logits stay f32) and a scale that is not a power of two take the exact-softmax variants."""
    q, k, v = _inputs(2, 4, 1024, seed=5)
    eng = _engine()
    for round_logits, scale in ((False, 0.125), (True, 0.11)):
        out = eng.attend(0, q, k, v, 4, scale, 1, round_logits, tapped=False)
        want, _ = _restated_f64(q, k, v, 4, scale, round_logits)
        assert (out.float().cpu() - want.float()).abs().max().item() <= 2e-3 * want.float().abs().max().item()
    eng.close()


def test_attend_declines_what_the_kernel_does_not_take():
    eng = _engine()
    q, k, v = _inputs(2, 4, 256, seed=1)
    assert eng.attend(0, q.float(), k.float(), v.float(), 4, 0.125, 1, True, tapped=False) is None        # fp32 pipeline
    assert eng.attend(0, q[:, :, :48].contiguous(), k[:, :, :48].contiguous(), v[:, :, :48].contiguous(), 4, 12 ** -0.5, 1, True,
                      tapped=False) is None                                                              # head_dim 12
    assert eng.attend(0, q[:, :, :128], k[:, :, :128], v[:, :, :128], 4, 32 ** -0.5, 1, True, tapped=False) is None  # strided views
    assert eng.attend(0, q, k[:, :64], v[:, :64], 4, 0.125, 1, True, tapped=False) is None                # 64 keys
    assert eng.attend(0, q.transpose(0, 1).contiguous().transpose(0, 1), k, v, 4, 0.125, 1, True, tapped=False) is None
    assert eng.attend(0, q, k, v, 4, 0.125, 1, True, tapped=False) is not None
    eng.close()


def test_extreme_logits_take_the_row_maximum_path():
    q, k, v = _inputs(2, 2, 256, seed=9, gain=6.0)         # logits of a few hundred: 2^((x - x0) L) overflows for some rows
    eng = _engine()
    out = eng.attend(0, q, k, v, 2, 0.125, 1, True, tapped=False)
    assert torch.isfinite(out).all()
    want, _ = _restated_f64(q, k, v, 2, 0.125)
    assert (out.float().cpu() - want.float()).abs().max().item() <= 4e-3 * want.float().abs().max().item()
    eng.close()


def test_processor_routes_leave_identical_sums(monkeypatch):
    """Through ``daam_amd.trace`` on an SDXL-shaped (head_dim 64, fp16) stack: deferred trace (daam_attend + batched tap),
    immediate trace (tap fused into daam_attend) and the framework-attention route (``DAAM_NO_ATTEND=1``: torch SDPA +
    batched tap) must leave bit-identical running sums (hence the same global heat maps); the hidden states of the two attention
    kernels agree to fp16 accuracy."""
    import daam_amd
    from oracle import fake_diffusers as fd
    pipe = fd.make_pipe('sdxl', device=DEV, dtype=torch.float16, batch=2, seed=41, mini=True, identity_proj=False,
                        dim_head=64, heads_scale=0.2, tblocks_cap=1)
    pipe.keep_outputs = True

    def run(**kw):
        with daam_amd.trace(pipe, **kw) as tc:
            pipe('a photo of a monkey', num_inference_steps=4)
            used = [h._attend is not None for h in tc.module if hasattr(h, '_attend')]
            raw = [(k, v.clone()) for k, v in tc.all_heat_maps]
            return raw, tc.compute_global_heat_map().heat_maps.clone(), [o.clone() for o in pipe.last_outputs], used

    deferred = run(defer_steps=64)
    immediate = run(defer_steps=0)
    monkeypatch.setenv('DAAM_NO_ATTEND', '1')
    stock = run(defer_steps=64)
    assert all(deferred[3]) and all(immediate[3]) and not any(stock[3])
    for other in (immediate, stock):
        assert [k for k, _ in other[0]] == [k for k, _ in deferred[0]]
        for (k, a), (_, b) in zip(deferred[0], other[0]):
            assert torch.equal(a, b), k
        # the finalize reduces its key chunks with f32 atomics: equal sums give equal maps up to the order of those adds
        assert (deferred[1] - other[1]).abs().max().item() <= 1e-6 * max(1.0, deferred[1].max().item())
    for a, b in zip(deferred[2], immediate[2]):
        assert torch.equal(a, b)
    for a, b in zip(deferred[2], stock[2]):
        assert (a.float() - b.float()).abs().max().item() <= 2e-3 * b.float().abs().max().item()


def test_cxx_recorder_runs_attend_in_the_steady_state():
    """On a deferred trace ``engine.attend`` IS the C++ recorder's entry point: after a layer's first call (which builds
    the descriptor in Python) the Python method is not entered again, the outputs are those of the Python path, and the
    recorded taps leave the same sums."""
    from daam_amd.engine import HeatMapEngine
    eng = HeatMapEngine(2, defer_steps=8)
    if eng._fast is None:
        pytest.skip('daam_amd._fastpath is not built')
    assert eng.attend == eng._fast.attend
    entered = []
    orig = HeatMapEngine.attend

    def counting(self, *a, **k):
        entered.append(a[0])
        return orig(self, *a, **k)
    HeatMapEngine.attend = counting
    try:
        ref = _engine(n_layers=2, defer_steps=0)           # immediate engine: Python attend with the fused tap
        outs, wants = [], []
        for step in range(4):
            for layer, (heads, hw) in enumerate(((10, 1024), (5, 4096))):
                q, k, v = _inputs(2, heads, hw, seed=10 * step + layer)
                outs.append(eng.attend(layer, q, k, v, heads, 0.125, 1, True, True))
                wants.append(orig(ref, layer, q, k, v, heads, 0.125, 1, True, True))
        assert entered == [0, 1]                           # one Python entry per layer, then C++ only
        assert eng.pending_taps == 8
        for a, b in zip(outs, wants):
            assert a is not None and torch.equal(a, b)
        got, want = dict(eng.items()), dict(ref.items())
        assert list(got) == list(want) and all(torch.equal(got[key], want[key]) for key in got)
        # a call the kernel does not take still reaches Python and declines
        q, k, v = _inputs(2, 10, 1024, seed=3)
        assert eng.attend(0, q.float(), k.float(), v.float(), 10, 0.125, 1, True, True) is None
    finally:
        HeatMapEngine.attend = orig
    eng.close()
    ref.close()

"""The attention processor itself (reference ``UNetCrossAttentionHooker.__call__``, trace.py:252-304) on the GPU: what
it RETURNS (the model's hidden states) and its less-travelled branches -- ``upcast_attention`` / ``upcast_softmax``,
attention masks, ``norm_cross``, ``low_memory``, the in-place guard and the stream hand-over of deferred taps.
Run with ``-m gpu`` on an MI355X."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import fake_diffusers as fd
from oracle import heatmap_oracle as ho
from oracle import torch_hooks as th

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'

REL = {torch.float32: 1e-5, torch.float16: 2e-3}


def _pipe(dtype, seed=31, **unet_kw):
    kw = dict(mini=True, identity_proj=False, dim_head=64, heads_scale=0.2, tblocks_cap=1)
    kw.update(unet_kw)
    pipe = fd.make_pipe('sdxl', device=DEV, dtype=dtype, batch=2, seed=seed, **kw)
    pipe.keep_outputs = True
    return pipe


def _run(pipe, steps=2, prompt='a photo of a monkey'):
    pipe(prompt, num_inference_steps=steps)
    return [o.clone() for o in pipe.last_outputs]


def _reference_run(pipe, steps=2, locate_kw=None, prompt='a photo of a monkey'):
    """The reference's processor on every attn2 (torch port), tapping the located ones: (outputs, RawMaps)."""
    import daam_amd
    modules = [s.module for s in pipe.unet.execution_order()]
    saved = [m.processor for m in modules]
    raw = th.RawMaps()
    located = daam_amd.UNetCrossAttentionLocator(**(locate_kw or {})).locate(pipe.unet)
    for m in modules:
        m.set_processor(th.ReferenceProcessor())
    for idx, m in enumerate(located):
        m.set_processor(th.ReferenceProcessor(raw, idx, 4096))
    try:
        outs = _run(pipe, steps, prompt)
    finally:
        for m, p in zip(modules, saved):
            m.set_processor(p)
    return outs, raw


def _assert_outputs(got, want, rel):
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape and a.dtype == b.dtype
        scale = b.float().abs().max().item()
        err = (a.float() - b.float()).abs().max().item()
        assert err <= rel * scale, f'attn2 call {i}: {err / scale:.2e} relative'


def _assert_maps(tc, raw, dtype, n_rows):
    got_items = list(tc.all_heat_maps)
    ref_items = list(raw)
    assert [k for k, _ in got_items] == [k for k, _ in ref_items]
    for (k, g), (_, w) in zip(got_items, ref_items):
        tol = 2e-6 if dtype == torch.float32 else 2.0 ** -10 * max(1.0, float(w.max()))
        assert (g.float() - w.float()).abs().max().item() <= tol, k
    want = th.global_heat_map(raw, 4096, n_rows=n_rows)
    got = tc.compute_global_heat_map().heat_maps
    assert (got - want).abs().max().item() <= (2e-6 if dtype == torch.float32 else 1e-3)


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
@pytest.mark.parametrize('route', ['qk', 'qk_immediate', 'probs'])
def test_hidden_states_match_reference_processor(dtype, route):
    """trace.py:296-304: the hooked processor must return what the reference's (= a stock materialising processor)
    returns -- fused route (SDPA) and materialised route, real projections."""
    import daam_amd
    pipe = _pipe(dtype)
    want_default = _run(pipe)                                           # DefaultProcessor (un-hooked)
    want, raw = _reference_run(pipe)
    _assert_outputs(want, want_default, 1e-6 if dtype == torch.float32 else 1e-3)   # the port IS a stock processor
    kw = dict(tap='probs') if route == 'probs' else dict(defer_steps=0 if route == 'qk_immediate' else 64)
    with daam_amd.trace(pipe, **kw) as tc:
        got = _run(pipe)
        _assert_outputs(got, want, REL[dtype])
        _assert_maps(tc, raw, dtype, len(pipe.tokenizer.tokenize('a photo of a monkey')) + 2)
    assert all(type(s.module.processor).__name__ == 'DefaultProcessor' for s in pipe.unet.execution_order())


@pytest.mark.parametrize('flag', ['upcast_attention', 'upcast_softmax', 'both'])
def test_upcast_flags(flag):
    """diffusers ``get_attention_scores`` with ``upcast_attention`` (f32 logits -> round_logits = 0 in the fused tap)
    and ``upcast_softmax`` (materialised route): maps and outputs against the reference's processor on fp16."""
    import daam_amd
    kw = dict(upcast_attention=flag in ('upcast_attention', 'both'), upcast_softmax=flag in ('upcast_softmax', 'both'))
    pipe = _pipe(torch.float16, seed=33, **kw)
    want, raw = _reference_run(pipe, steps=3)
    with daam_amd.trace(pipe) as tc:
        got = _run(pipe, steps=3)
        hookers = [h for h in tc.module if hasattr(h, '_fusable')]
        assert all(h._fusable == (not kw['upcast_softmax']) for h in hookers)
        assert all(h._round_logits == (not kw['upcast_attention']) for h in hookers)
        _assert_outputs(got, want, REL[torch.float16])
        _assert_maps(tc, raw, torch.float16, 7)
    # the same inputs WITHOUT the flag give different logits roundings: the flag is really honoured
    if flag == 'upcast_attention':
        cpu = fd.make_pipe('sdxl', device='cpu', dtype=torch.float16, batch=2, seed=33, mini=True, identity_proj=False,
                           dim_head=64, heads_scale=0.2, tblocks_cap=1, upcast_attention=True)
        with torch.no_grad():
            a = ho.replay_generation(cpu, 1, torch.float16)
            for s in cpu.unet.execution_order():
                s.module.upcast_attention = False
            b = ho.replay_generation(cpu, 1, torch.float16)
        assert any((x != y).any() for (_, x), (_, y) in zip(a, b))


class _KeyLengthMaskAttention(fd.FakeAttention):
    """``prepare_attention_mask`` as later diffusers releases call it (target = KEY length): a ``[B, 1, 77]`` additive
    bias reaches ``get_attention_scores`` as ``[B*H, 1, 77]``.  (With the 0.21.2 call convention the reference pads
    the mask to 77 + hw columns and fails, see test_attention_mask_0_21_2_semantics.)"""

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None, out_dim=3):
        if attention_mask is None:
            return None
        return attention_mask.repeat_interleave(self.heads, dim=0) if attention_mask.shape[0] < batch_size * self.heads \
            else attention_mask


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_attention_mask_goes_through_the_materialised_route(dtype):
    import daam_amd
    pipe = _pipe(dtype, seed=35)
    for s in pipe.unet.execution_order():
        s.module.__class__ = _KeyLengthMaskAttention
    bias = torch.zeros(2, 1, 77, device=DEV, dtype=dtype)
    bias[:, :, 40:] = -10000.0                                           # padding tokens masked out
    bias[1, :, 3] = -4.0
    pipe.mask_fn = lambda i, spec: bias
    want, raw = _reference_run(pipe)
    with daam_amd.trace(pipe) as tc:
        got = _run(pipe)
        _assert_outputs(got, want, REL[dtype])
        _assert_maps(tc, raw, dtype, 7)
        # masked tokens carry no attention
        for _, v in tc.all_heat_maps:
            assert v[40:].abs().max().item() == 0.0
    # ... and the numpy oracle agrees on what a mask means
    a = pipe.unet.execution_order()[0]
    hs, ctx = pipe.hidden_states(0, a, 0), pipe.context(0, a)
    with torch.no_grad():
        qt = a.module.head_to_batch_dim(a.module.to_q(hs))
        kt = a.module.head_to_batch_dim(a.module.to_k(ctx))
        t = a.module.get_attention_scores(qt, kt, bias.repeat_interleave(a.heads, 0))
    q, k = qt.float().cpu().numpy(), kt.float().cpu().numpy()
    np_dt = np.float32 if dtype == torch.float32 else np.float16
    m = bias.repeat_interleave(a.heads, 0).float().cpu().numpy()
    p = ho.attention_probs(q.astype(np_dt), k.astype(np_dt), a.module.scale, np_dt, mask=m)
    assert np.abs(p.astype(np.float32) - t.float().cpu().numpy()).max() <= (1e-6 if dtype == torch.float32 else 2.0 ** -10)


def test_attention_mask_0_21_2_semantics():
    """diffusers 0.21.2 + the reference's call ``prepare_attention_mask(mask, QUERY length, batch)`` (trace.py:259-260):
    a [B, 1, 77] cross-attention bias is padded by hw zero columns and ``baddbmm`` rejects it -- the reference raises
    RuntimeError, and so does the drop-in (same call, same route)."""
    import daam_amd
    pipe = _pipe(torch.float16, seed=36)
    pipe.mask_fn = lambda i, spec: torch.zeros(2, 1, 77, device=DEV, dtype=torch.float16)
    with pytest.raises(RuntimeError):
        _reference_run(pipe, steps=1)
    with daam_amd.trace(pipe):
        with pytest.raises(RuntimeError):
            _run(pipe, steps=1)


def test_norm_cross():
    """trace.py:264-267: ``attn.norm_cross`` is applied to the encoder states before to_k / to_v."""
    import daam_amd
    pipe = _pipe(torch.float16, seed=37)
    torch.manual_seed(5)
    for s in pipe.unet.execution_order():
        ln = nn.LayerNorm(s.module.to_v.in_features).to(device=DEV, dtype=torch.float16)
        with torch.no_grad():
            ln.weight.uniform_(0.5, 2.0)
            ln.bias.uniform_(-0.5, 0.5)
        s.module.norm_cross = ln
    want, raw = _reference_run(pipe)
    with daam_amd.trace(pipe) as tc:
        got = _run(pipe)
        _assert_outputs(got, want, REL[torch.float16])
        _assert_maps(tc, raw, torch.float16, 7)


def test_low_memory_restricts_to_first_attention_of_each_block():
    """trace(pipe, low_memory=True) hooks ``restrict={0}`` (trace.py:35): keys, maps and layer names."""
    import daam_amd
    pipe = _pipe(torch.float16, seed=38, tblocks_cap=2)
    want, raw = _reference_run(pipe, locate_kw=dict(restrict={0}))
    with daam_amd.trace(pipe, low_memory=True) as tc:
        got = _run(pipe)
        assert len(tc.layer_names) == 4 and tc.engine.n_layers == 4     # up0, up1, down1, down2: one attn2 each
        _assert_outputs(got, want, REL[torch.float16])
        _assert_maps(tc, raw, torch.float16, 7)
        assert tc.all_heat_maps.layers() == {0, 1, 2, 3}


def test_deferred_tap_detects_in_place_writes(monkeypatch):
    """Deferred taps read Q / K when the launch is issued: ``DAAM_CHECK_VERSIONS=1`` turns a forbidden in-place write
    between the processor call and the launch into an error instead of a silently different heat map."""
    from daam_amd.engine import HeatMapEngine
    monkeypatch.setenv('DAAM_CHECK_VERSIONS', '1')
    q = torch.randn(2, 256, 128, device=DEV, dtype=torch.float16)
    k = torch.randn(2, 77, 128, device=DEV, dtype=torch.float16)
    eng = HeatMapEngine(1, defer_steps=8)
    eng.tap_qk(0, q, k, 2, 0.125, factor=4)
    eng.flush()                                                          # untouched: fine
    eng.tap_qk(0, q, k, 2, 0.125, factor=4)
    q.mul_(2.0)
    with pytest.raises(RuntimeError, match='modified in place'):
        eng.flush()
    eng.close()


def test_deferred_taps_follow_the_recording_stream():
    """Q / K are produced on a side stream; the maps are read from the default stream.  The flush must be ordered
    after the producers (and the producers' memory must not be recycled under the tap kernel)."""
    from daam_amd.engine import HeatMapEngine
    heads, d, hw, steps = 4, 64, 1024, 6
    g = torch.Generator(device=DEV).manual_seed(3)
    k = torch.randn(2, 77, heads * d, generator=g, device=DEV, dtype=torch.float16)
    base = [torch.randn(2, hw, heads * d, generator=g, device=DEV, dtype=torch.float16) for _ in range(steps)]
    ref = HeatMapEngine(1, defer_steps=0)
    for q in base:
        ref.tap_qk(0, q * 1.0, k, heads, d ** -0.5, factor=2)
    want = torch.stack([v for _, v in ref.items()]).clone()
    ref.close()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=DEV)
    eng = HeatMapEngine(1, defer_steps=64)
    big = torch.empty(64 << 20, device=DEV, dtype=torch.float16)
    with torch.cuda.stream(side):
        for q in base:
            for _ in range(4):
                big.normal_()                                            # keep the side stream busy ahead of the producer
            eng.tap_qk(0, q * 1.0, k, heads, d ** -0.5, factor=2)        # Q produced on the side stream
    got = torch.stack([v for _, v in eng.items()])                        # flush + read on the default stream
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    eng.close()


def test_hooked_trace_stays_alive_without_a_reference():
    """``trace(pipe).hook()`` with the object dropped is legal with the reference (its hookers hold the trace strongly):
    the installed processors must keep working, and unhooking releases the pin again."""
    import gc
    import weakref
    import daam_amd
    pipe = _pipe(torch.float16, seed=39)
    daam_amd.trace(pipe).hook()
    gc.collect()
    _run(pipe)                                                           # no ReferenceError / NoneType errors
    located = daam_amd.UNetCrossAttentionLocator().locate(pipe.unet)
    tc = located[0].processor._pinned_trace
    assert tc.compute_global_heat_map().heat_maps.shape == (7, 64, 64)
    ref = weakref.ref(tc)
    tc.unhook()
    del tc
    gc.collect()
    assert ref() is None                                                 # no cycle left behind
    assert type(located[0].processor).__name__ == 'DefaultProcessor'

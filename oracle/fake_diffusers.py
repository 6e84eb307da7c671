"""TEST INFRASTRUCTURE (oracle side) -- not part of the shipped product path.

A minimal stand-in for the pieces of ``diffusers==0.21.2`` that the DAAM hot path
touches (``requirements.txt:2`` of the reference pins that release; it is NOT
installed and not installable in this image).  It serves two purposes:

1. ``install_stubs()`` registers fake ``diffusers`` / ``spacy`` modules so that the
   *unmodified* reference package under ``/root/reference`` can be imported and
   executed in this container.  That is how the golden
   vectors in ``tests/golden/`` were produced (``oracle/make_golden.py``).
2. The same fake UNet / pipeline objects are what the parity tests hook
   ``daam_amd.trace`` onto: ``daam_amd`` duck-types the attention-processor protocol
   and never imports ``diffusers`` itself.

``FakeAttention`` restates ``diffusers.models.attention_processor.Attention`` as used at
reference ``daam/trace.py:261-302``: bias-free ``to_q/to_k/to_v``,
``to_out = [Linear, Dropout]``, ``heads``, ``scale = dim_head ** -0.5``,
``head_to_batch_dim`` (batch-major / head-minor), ``batch_to_head_dim``,
``prepare_attention_mask`` and ``get_attention_scores`` (= ``baddbmm(alpha=scale)`` ->
``softmax(-1)`` -> cast back to the query dtype).

UNet topologies (block structure, attention counts, transformer blocks per attention,
head counts, resolutions) follow SD-v1.5 and SDXL-base; channel widths / head dims can
be shrunk (``mini=True``) so CPU runs of the reference finish in seconds.
"""
from __future__ import annotations

import math
import sys
import types
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.nn as nn


# --------------------------------------------------------------------------------------
# Attention (diffusers 0.21.2 restatement)
# --------------------------------------------------------------------------------------
class _Passthrough(nn.Module):
    """Identity projection: lets a test feed pre-projected Q / K bits straight through."""

    def forward(self, x):
        return x


class FakeAttention(nn.Module):
    def __init__(self, query_dim: int, cross_attention_dim: int, heads: int, dim_head: int,
                 identity_proj: bool = False, upcast_attention: bool = False,
                 upcast_softmax: bool = False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.dim_head = dim_head
        self.inner_dim = inner
        self.scale = dim_head ** -0.5
        self.upcast_attention = upcast_attention
        self.upcast_softmax = upcast_softmax
        self.norm_cross = None
        self.group_norm = None
        if identity_proj:
            if query_dim != inner or cross_attention_dim != inner:
                raise ValueError('identity projections need query_dim == cross_dim == heads*dim_head')
            self.to_q, self.to_k = _Passthrough(), _Passthrough()
        else:
            self.to_q = nn.Linear(query_dim, inner, bias=False)
            self.to_k = nn.Linear(cross_attention_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_attention_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.processor = DefaultProcessor()

    # -- protocol used by the reference at trace.py:261-311 --------------------------
    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None, out_dim=3):
        """diffusers 0.21.2 ``Attention.prepare_attention_mask``: ``None`` passes through; a mask whose last dim
        differs from ``target_length`` is padded by ``target_length`` zeros (sic); then repeated per head.  The
        reference passes the QUERY length as ``target_length`` (trace.py:259-260), so a ``[B, 1, 77]`` cross-attention
        bias is padded to ``77 + hw`` columns and ``baddbmm`` then fails -- a masked cross-attention call is an
        error in the reference unless the mask is already ``hw`` columns wide."""
        head_size = self.heads
        if attention_mask is None:
            return attention_mask
        current_length = attention_mask.shape[-1]
        if current_length != target_length:
            attention_mask = torch.nn.functional.pad(attention_mask, (0, target_length), value=0.0)
        if out_dim == 3:
            if attention_mask.shape[0] < batch_size * head_size:
                attention_mask = attention_mask.repeat_interleave(head_size, dim=0)
        elif out_dim == 4:
            attention_mask = attention_mask.unsqueeze(1)
            attention_mask = attention_mask.repeat_interleave(head_size, dim=1)
        return attention_mask

    def head_to_batch_dim(self, t):
        b, s, c = t.shape
        h = self.heads
        return t.reshape(b, s, h, c // h).permute(0, 2, 1, 3).reshape(b * h, s, c // h)

    def batch_to_head_dim(self, t):
        bh, s, d = t.shape
        h = self.heads
        return t.reshape(bh // h, h, s, d).permute(0, 2, 1, 3).reshape(bh // h, s, d * h)

    def get_attention_scores(self, query, key, attention_mask=None):
        dtype = query.dtype
        if self.upcast_attention:
            query, key = query.float(), key.float()
        if attention_mask is None:
            base = torch.empty(query.shape[0], query.shape[1], key.shape[1],
                               dtype=query.dtype, device=query.device)
            beta = 0
        else:
            base, beta = attention_mask, 1
        scores = torch.baddbmm(base, query, key.transpose(-1, -2), beta=beta, alpha=self.scale)
        if self.upcast_softmax:
            scores = scores.float()
        probs = scores.softmax(dim=-1)
        return probs.to(dtype)


class DefaultProcessor:
    """What a stock pipeline would run when DAAM is not hooked (diffusers 0.21.2 ``AttnProcessor``: materialised
    attention; the mask is prepared for the KEY length, ``norm_cross`` is applied to the encoder states)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        batch, key_length, _ = ctx.shape
        attention_mask = attn.prepare_attention_mask(attention_mask, key_length, batch)
        if encoder_hidden_states is not None and attn.norm_cross is not None:
            ctx = attn.norm_cross(ctx)
        q = attn.head_to_batch_dim(attn.to_q(hidden_states))
        k = attn.head_to_batch_dim(attn.to_k(ctx))
        v = attn.head_to_batch_dim(attn.to_v(ctx))
        probs = attn.get_attention_scores(q, k, attention_mask)
        out = attn.batch_to_head_dim(torch.bmm(probs, v))
        return attn.to_out[1](attn.to_out[0](out))


# --------------------------------------------------------------------------------------
# UNet topology
# --------------------------------------------------------------------------------------
class _TransformerBlock(nn.Module):
    def __init__(self, attn2: FakeAttention):
        super().__init__()
        self.attn2 = attn2


class _Transformer2D(nn.Module):
    def __init__(self, blocks: Sequence[_TransformerBlock]):
        super().__init__()
        self.transformer_blocks = nn.ModuleList(blocks)


class _AttnBlockBase(nn.Module):
    def __init__(self, res: int, n_attn: int, n_tblocks: int, make_attn):
        super().__init__()
        self.res = res
        self.attentions = nn.ModuleList([
            _Transformer2D([_TransformerBlock(make_attn()) for _ in range(n_tblocks)])
            for _ in range(n_attn)])


# The reference locator keys on ``'CrossAttn' in block.__class__.__name__`` (hook.py:115).
class CrossAttnDownBlock2D(_AttnBlockBase):
    pass


class CrossAttnUpBlock2D(_AttnBlockBase):
    pass


class UNetMidBlock2DCrossAttn(_AttnBlockBase):
    pass


class DownBlock2D(nn.Module):
    pass


class UpBlock2D(nn.Module):
    pass


@dataclass
class UNetConfig:
    sample_size: int


@dataclass
class LayerSpec:
    """One hooked ``attn2`` in *execution* order (down -> mid -> up)."""
    module: FakeAttention
    res: int           # side of the square feature map
    heads: int
    dim_head: int
    query_dim: int


class FakeUNet(nn.Module):
    """Only the structure the reference locator walks (hook.py:105-127) plus a forward
    that drives every cross-attention once per denoising step with caller-supplied
    hidden states."""

    def __init__(self, kind: str, *, sample_size: Optional[int] = None, mini: bool = True,
                 identity_proj: bool = True, heads_scale: float = 1.0, tblocks_cap: Optional[int] = None,
                 dim_head: Optional[int] = None, latent_size: Optional[int] = None,
                 upcast_attention: bool = False, upcast_softmax: bool = False):
        super().__init__()
        self.kind = kind
        if kind == 'sd15':
            sample = 64 if sample_size is None else sample_size
            chans = (320, 640, 1280, 1280)
            heads = [8, 8, 8, 8]
            dheads = [c // 8 for c in chans]
            cross_dim = 768
            down_plan = [('x', 2, 1), ('x', 2, 1), ('x', 2, 1), (None, 0, 0)]
            up_plan = [(None, 0, 0), ('x', 3, 1), ('x', 3, 1), ('x', 3, 1)]
            mid_tblocks = 1
        elif kind == 'sdxl':
            sample = 128 if sample_size is None else sample_size
            chans = (320, 640, 1280)
            heads = [5, 10, 20]
            dheads = [64, 64, 64]
            cross_dim = 2048
            down_plan = [(None, 0, 0), ('x', 2, 2), ('x', 2, 10)]
            up_plan = [('x', 3, 10), ('x', 3, 2), (None, 0, 0)]
            mid_tblocks = 10
        else:
            raise ValueError(kind)
        self.config = UNetConfig(sample_size=sample)
        self.cross_dim = cross_dim

        def scaled_heads(h):
            return max(1, int(round(h * heads_scale)))

        def level_params(level):
            h = scaled_heads(heads[level])
            d = dheads[level] if dim_head is None else dim_head
            if mini and dim_head is None:
                d = max(8, d // 8)
            return h, d

        def make_factory(level):
            h, d = level_params(level)
            inner = h * d
            flags = dict(upcast_attention=upcast_attention, upcast_softmax=upcast_softmax)
            if identity_proj:
                return lambda: FakeAttention(inner, inner, h, d, identity_proj=True, **flags)
            qd = inner if mini else chans[level]
            cd = inner if mini else cross_dim
            return lambda: FakeAttention(qd, cd, h, d, **flags)

        def cap(n):
            return n if tblocks_cap is None else min(n, tblocks_cap)

        # ``config.sample_size`` is fixed at model-load time; a pipeline asked for a larger
        # image (SDXL at 2048x2048) runs the same UNet on a larger latent (``latent_size``).
        n_levels = len(chans)
        downs, res = [], (sample if latent_size is None else latent_size)
        self._down_res = []
        for level, (kind_, n_attn, n_tb) in enumerate(down_plan):
            self._down_res.append(res)
            if kind_ is None:
                downs.append(DownBlock2D())
            else:
                downs.append(CrossAttnDownBlock2D(res, n_attn, cap(n_tb), make_factory(level)))
            if level != n_levels - 1:
                res //= 2
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = UNetMidBlock2DCrossAttn(res, 1, cap(mid_tblocks), make_factory(n_levels - 1))
        ups = []
        for i, (kind_, n_attn, n_tb) in enumerate(up_plan):
            level = n_levels - 1 - i
            if kind_ is None:
                ups.append(UpBlock2D())
            else:
                ups.append(CrossAttnUpBlock2D(res, n_attn, cap(n_tb), make_factory(level)))
            if i != n_levels - 1:
                res *= 2
        self.up_blocks = nn.ModuleList(ups)

    # execution order of every cross-attention (down -> mid -> up), as a UNet forward visits them
    def execution_order(self, include_mid: bool = True) -> List[LayerSpec]:
        out: List[LayerSpec] = []
        blocks = list(self.down_blocks) + ([self.mid_block] if include_mid else []) + list(self.up_blocks)
        for blk in blocks:
            if not isinstance(blk, _AttnBlockBase):
                continue
            for tr in blk.attentions:
                for tb in tr.transformer_blocks:
                    a = tb.attn2
                    qd = a.to_out[0].out_features
                    out.append(LayerSpec(a, blk.res, a.heads, a.dim_head, qd))
        return out

    def forward(self, hidden_fn, context_fn, step: int, mask_fn=None):
        """``hidden_fn(spec_index, spec, step) -> [B, res*res, query_dim]`` and
        ``context_fn(spec_index, spec) -> [B, 77, cross_dim_of_layer]``; optional
        ``mask_fn(spec_index, spec) -> attention_mask`` (stock SD / SDXL pipelines pass none)."""
        outs = []
        for i, spec in enumerate(self.execution_order()):
            hs = hidden_fn(i, spec, step)
            ctx = context_fn(i, spec)
            if mask_fn is None:
                outs.append(spec.module(hs, encoder_hidden_states=ctx))
            else:
                outs.append(spec.module(hs, encoder_hidden_states=ctx, attention_mask=mask_fn(i, spec)))
        return outs


# --------------------------------------------------------------------------------------
# Pipeline
# --------------------------------------------------------------------------------------
class FakeTokenizer:
    """``tokenize`` splits on whitespace; a word longer than 6 chars becomes two sub-word
    pieces so ``compute_token_merge_indices`` has something to merge.  Suffix ``</w>``
    mimics the CLIP BPE end-of-word marker (reference utils.py:76)."""
    model_max_length = 77

    def tokenize(self, text: str) -> List[str]:
        toks: List[str] = []
        for w in text.split():
            if len(w) > 6:
                toks += [w[:4], w[4:] + '</w>']
            else:
                toks.append(w + '</w>')
        return toks


class FakeImageProcessor:
    def postprocess(self, image, output_type='pil', **kw):
        return [image] if not isinstance(image, list) else image

    def numpy_to_pil(self, image):
        return [image] if not isinstance(image, list) else image


class _PipeBase:
    def __init__(self, unet: FakeUNet, device='cpu', dtype=torch.float32, batch: int = 2,
                 sos_gain: float = 3.0):
        self.unet = unet
        self.vae_scale_factor = 8
        self.tokenizer = FakeTokenizer()
        self.image_processor = FakeImageProcessor()
        self.device = torch.device(device)
        self.dtype = dtype
        self.batch = batch            # 2 = classifier-free guidance [uncond, cond]
        self.sos_gain = sos_gain
        self.seed = 0
        self.checked = []
        self.mask_fn = None           # optional: (i, spec) -> attention_mask for every cross-attention call
        self.keep_outputs = False     # True: ``last_outputs`` = what every attn2 returned in the final step
        self.last_outputs = None

    # patched by PipelineHooker (trace.py:171-186)
    def check_inputs(self, prompt, *args, **kwargs):
        self.checked.append(prompt)

    # ---- synthetic inputs: deterministic in (seed, layer, step) -----------------------
    def _gen(self, *key):
        g = torch.Generator(device='cpu')
        mix = self.seed & 0xFFFF
        for v in key:
            mix = (mix * 1000003 + int(v) + 12345) % (2 ** 31 - 1)
        g.manual_seed(mix)
        return g

    def hidden_states(self, i, spec: LayerSpec, step: int):
        g = self._gen(1, i, step)
        x = torch.randn(self.batch, spec.res * spec.res, spec.query_dim, generator=g)
        return x.to(self.dtype).to(self.device)

    def context(self, i, spec: LayerSpec):
        cd = spec.module.to_v.in_features
        g = self._gen(2, i)
        c = torch.randn(self.batch, 77, cd, generator=g)
        c[:, 0, :] *= self.sos_gain          # SOS-dominant, like real maps
        return c.to(self.dtype).to(self.device)

    def __call__(self, prompt, num_inference_steps=5, generator=None, callback=None, **kw):
        self.check_inputs(prompt, 512, 512, 1)
        with torch.no_grad():
            for step in range(num_inference_steps):
                outs = self.unet(self.hidden_states, self.context, step, self.mask_fn)
                if self.keep_outputs and step == num_inference_steps - 1:
                    self.last_outputs = outs
                if callback is not None:
                    callback(step, step, None)
        image = 'image-of:' + (prompt if isinstance(prompt, str) else prompt[0])
        return self._finish(image)

    def _finish(self, image):
        raise NotImplementedError


class StableDiffusionPipeline(_PipeBase):
    def run_safety_checker(self, image, device=None, dtype=None):
        return image, None

    def _finish(self, image):
        image, _ = self.run_safety_checker(image, None, None)
        return types.SimpleNamespace(images=[image])


class StableDiffusionXLPipeline(_PipeBase):
    def _finish(self, image):
        images = self.image_processor.postprocess(image, output_type='pil')
        return types.SimpleNamespace(images=images)


class DiffusionPipeline(_PipeBase):
    pass


def make_pipe(kind: str, *, device='cpu', dtype=torch.float32, batch=2, seed=0, **unet_kw):
    # the to_v / to_out weights (the only random parameters; they do not influence any heat map) are drawn on the
    # CPU from a generator state fixed by ``seed``, so the processors' outputs are reproducible bit for bit
    rng_state = torch.get_rng_state()
    torch.manual_seed(1000 + seed)
    try:
        unet = FakeUNet(kind, **unet_kw)
    finally:
        torch.set_rng_state(rng_state)
    unet = unet.to(device=device, dtype=dtype)
    cls = StableDiffusionXLPipeline if kind == 'sdxl' else StableDiffusionPipeline
    pipe = cls(unet, device=device, dtype=dtype, batch=batch)
    pipe.seed = seed
    return pipe


# --------------------------------------------------------------------------------------
# sys.modules stubs so the real reference imports (c2)
# --------------------------------------------------------------------------------------
def install_stubs():
    me = sys.modules[__name__]
    if 'diffusers' in sys.modules and getattr(sys.modules['diffusers'], '_daam_fake', False):
        return
    d = types.ModuleType('diffusers')
    d._daam_fake = True
    d.StableDiffusionPipeline = StableDiffusionPipeline
    d.StableDiffusionXLPipeline = StableDiffusionXLPipeline
    d.DiffusionPipeline = DiffusionPipeline
    d.UNet2DConditionModel = FakeUNet
    ip = types.ModuleType('diffusers.image_processor')
    ip.VaeImageProcessor = FakeImageProcessor
    models = types.ModuleType('diffusers.models')
    ap = types.ModuleType('diffusers.models.attention_processor')
    ap.Attention = FakeAttention
    models.attention_processor = ap
    d.image_processor, d.models = ip, models
    sys.modules.update({'diffusers': d, 'diffusers.image_processor': ip,
                        'diffusers.models': models, 'diffusers.models.attention_processor': ap})
    sp = types.ModuleType('spacy')
    spt = types.ModuleType('spacy.tokens')

    class Token:  # used at class-definition time by reference heatmap.py:111
        pass

    spt.Token = Token
    sp.tokens = spt
    sp.load = lambda *a, **k: (_ for _ in ()).throw(OSError('spacy is a stub'))
    sys.modules.update({'spacy': sp, 'spacy.tokens': spt})


def import_reference(path: str = '/root/reference'):
    """Import the unmodified reference package.  Returns ``(package, trace_module)``;
    ``daam.trace`` the *name* resolves to the class (reference trace.py:318), so the module
    object is taken from ``sys.modules``."""
    install_stubs()
    if path not in sys.path:
        sys.path.insert(0, path)
    import matplotlib
    matplotlib.use('Agg')
    import daam  # noqa: F401
    return sys.modules['daam'], sys.modules['daam.trace']

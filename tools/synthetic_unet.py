"""A synthetic cross-attention stack with SDXL-base / SD-v1.5 geometry for ``bench.py``'s integrated-overhead leg.

It is a WORKLOAD, not a checker (nothing here computes a heat map): ``torch.nn.Linear`` projections of the real widths,
the attribute / method surface of ``diffusers.models.attention_processor.Attention`` that an attention processor touches
(``to_q / to_k / to_v / to_out / heads / scale / norm_cross / set_processor / head_to_batch_dim / batch_to_head_dim /
prepare_attention_mask / get_attention_scores``), UNet blocks laid out the way ``UNetCrossAttentionLocator`` walks them,
and a pipeline-shaped driver that calls every ``attn2`` once per denoising step with device-resident hidden states.
The extraction overhead of a denoising step = step time with ``daam_amd.trace`` hooked minus step time with the stock
fused-SDPA processor below (metric (i), integrated harness).
"""
from __future__ import annotations

import types
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F


class SdpaProcessor:
    """What a stock pipeline runs on an ``attn2`` (AttnProcessor2_0 reduced to its tensor ops)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **_):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        b = hidden_states.shape[0]
        q, k, v = attn.to_q(hidden_states), attn.to_k(ctx), attn.to_v(ctx)
        d = q.shape[-1] // attn.heads
        q, k, v = (t.view(b, -1, attn.heads, d).transpose(1, 2) for t in (q, k, v))
        out = F.scaled_dot_product_attention(q, k, v, scale=attn.scale)
        out = out.transpose(1, 2).reshape(b, -1, attn.heads * d)
        return attn.to_out[1](attn.to_out[0](out))


class Attention(nn.Module):
    def __init__(self, query_dim: int, cross_dim: int, heads: int, dim_head: int):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.scale = heads, dim_head ** -0.5
        self.upcast_attention = self.upcast_softmax = False
        self.norm_cross = None
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(cross_dim, inner, bias=False)
        self.to_v = nn.Linear(cross_dim, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])
        self.processor = SdpaProcessor()

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kw):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, **kw)

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None, out_dim=3):
        return attention_mask

    def head_to_batch_dim(self, t):
        b, s, c = t.shape
        return t.reshape(b, s, self.heads, c // self.heads).permute(0, 2, 1, 3).reshape(b * self.heads, s, c // self.heads)

    def batch_to_head_dim(self, t):
        bh, s, d = t.shape
        return t.reshape(bh // self.heads, self.heads, s, d).permute(0, 2, 1, 3).reshape(bh // self.heads, s, d * self.heads)

    def get_attention_scores(self, query, key, attention_mask=None):
        scores = torch.baddbmm(torch.empty(query.shape[0], query.shape[1], key.shape[1], dtype=query.dtype, device=query.device),
                               query, key.transpose(-1, -2), beta=0, alpha=self.scale)
        return scores.softmax(dim=-1).to(query.dtype)


class _TransformerBlock(nn.Module):
    def __init__(self, attn2):
        super().__init__()
        self.attn2 = attn2


class _Transformer2D(nn.Module):
    def __init__(self, n_tblocks, make):
        super().__init__()
        self.transformer_blocks = nn.ModuleList([_TransformerBlock(make()) for _ in range(n_tblocks)])


class _Block(nn.Module):
    def __init__(self, res, n_attn, n_tblocks, make):
        super().__init__()
        self.res = res
        self.attentions = nn.ModuleList([_Transformer2D(n_tblocks, make) for _ in range(n_attn)])


class CrossAttnDownBlock2D(_Block):
    pass


class CrossAttnUpBlock2D(_Block):
    pass


class UNetMidBlock2DCrossAttn(_Block):
    pass


class PlainBlock(nn.Module):
    pass


class SyntheticUNet(nn.Module):
    """SDXL-base: channels (320, 640, 1280), heads 5 / 10 / 20 (head_dim 64), transformer blocks 1 / 2 / 10,
    cross-attention in down 1-2, mid, up 0-1, context width 2048.  SD-v1.5: (320, 640, 1280, 1280), 8 heads."""

    def __init__(self, kind: str = 'sdxl', latent: int = 128):
        super().__init__()
        if kind == 'sdxl':
            chans, heads, cross = (320, 640, 1280), (5, 10, 20), 2048
            down = [(None, 0, 0), ('x', 2, 2), ('x', 2, 10)]
            up = [('x', 3, 10), ('x', 3, 2), (None, 0, 0)]
            mid_tb, sample = 10, 128
        else:
            chans, heads, cross = (320, 640, 1280, 1280), (8, 8, 8, 8), 768
            down = [('x', 2, 1), ('x', 2, 1), ('x', 2, 1), (None, 0, 0)]
            up = [(None, 0, 0), ('x', 3, 1), ('x', 3, 1), ('x', 3, 1)]
            mid_tb, sample = 1, 64
        self.config = types.SimpleNamespace(sample_size=sample)
        n = len(chans)

        def maker(level):
            return lambda: Attention(chans[level], cross, heads[level], chans[level] // heads[level])
        res, downs = latent, []
        for level, (k, na, nt) in enumerate(down):
            downs.append(PlainBlock() if k is None else CrossAttnDownBlock2D(res, na, nt, maker(level)))
            if level != n - 1:
                res //= 2
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = UNetMidBlock2DCrossAttn(res, 1, mid_tb, maker(n - 1))
        ups = []
        for i, (k, na, nt) in enumerate(up):
            ups.append(PlainBlock() if k is None else CrossAttnUpBlock2D(res, na, nt, maker(n - 1 - i)))
            if i != n - 1:
                res *= 2
        self.up_blocks = nn.ModuleList(ups)

    def attn2_in_execution_order(self) -> List[tuple]:
        out = []
        for blk in list(self.down_blocks) + [self.mid_block] + list(self.up_blocks):
            if isinstance(blk, _Block):
                out += [(tb.attn2, blk.res) for tr in blk.attentions for tb in tr.transformer_blocks]
        return out


class SyntheticPipeline:
    """``pipe(prompt, num_inference_steps)``: every attn2 once per step, hidden states resident on the device."""

    def __init__(self, kind='sdxl', latent=128, device='cuda:0', dtype=torch.float16, n_sets=4, gain=3.0, seed=7):
        torch.manual_seed(seed)
        self.unet = SyntheticUNet(kind, latent).to(device=device, dtype=dtype)
        self.vae_scale_factor = 8
        self.tokenizer = types.SimpleNamespace(tokenize=lambda text: [w + '</w>' for w in text.split()])
        self.image_processor = types.SimpleNamespace(postprocess=lambda image, output_type='pil', **kw: [image])
        self.order = self.unet.attn2_in_execution_order()
        g = torch.Generator(device=device).manual_seed(seed)
        self.hidden = [[torch.randn(2, res * res, a.to_q.in_features, generator=g, device=device, dtype=dtype) * gain
                        for a, res in self.order] for _ in range(n_sets)]
        self.context = []
        for a, _ in self.order:
            c = torch.randn(2, 77, a.to_k.in_features, generator=g, device=device, dtype=dtype) * gain
            c[:, 0] *= 3.0
            self.context.append(c)

    def check_inputs(self, prompt, *args, **kwargs):
        pass

    def __call__(self, prompt, num_inference_steps=50, **kw):
        self.check_inputs(prompt, 1024, 1024, 1)
        with torch.no_grad():
            for step in range(num_inference_steps):
                hs = self.hidden[step % len(self.hidden)]
                for i, (attn, _) in enumerate(self.order):
                    attn(hs[i], encoder_hidden_states=self.context[i])
        return types.SimpleNamespace(images=self.image_processor.postprocess('synthetic'))


SyntheticPipeline.__name__ = 'StableDiffusionXLPipeline'     # trace() hooks image_processor for SDXL pipelines (trace.py:55-56)

"""``trace`` / ``DiffusionHeatMapHooker``: the reference's tracing API (``daam/trace.py``)
over the MI355X-native extraction path.

Differences from the reference are confined to *how* the per-layer work is done:
  * the cross-attention processor does not materialise ``attention_probs`` on the default
    path: the model's output comes from fused SDPA, and the heat-map tap recomputes the
    conditional-half probabilities from the projected Q / K inside one HIP kernel
    (``daam_tap_qk``), optionally deferred so that several denoising steps of all layers
    run as a single launch;
  * ``compute_global_heat_map`` is one fused bicubic + clamp + mean kernel.
The materialised path (``get_attention_scores`` -> ``daam_tap_probs`` -> ``bmm``) is kept for
``save_heads`` / ``load_heads``, attention masks, and ``tap='probs'``.
"""
from __future__ import annotations

import math
import os
from pathlib import Path
from typing import Any, List, Optional, Type, Union

import torch
import torch.nn.functional as F

from .engine import HeatMapEngine
from .heatmap import GlobalHeatMap, RawHeatMapCollection
from .hook import AggregateHooker, ObjectHooker, UNetCrossAttentionLocator
from .utils import cache_dir

__all__ = ['trace', 'DiffusionHeatMapHooker', 'GlobalHeatMap']


def _default_defer() -> int:
    return int(os.environ.get('DAAM_DEFER_STEPS', '64'))


def _default_defer_bytes(pipeline=None) -> int:
    """Bytes of recorded Q / K a trace may keep alive between tap launches: ``$DAAM_DEFER_BYTES``, else
    32 GiB, but never more than a quarter of the device memory that is free when the trace is set up."""
    env = os.environ.get('DAAM_DEFER_BYTES')
    if env:
        return int(env)
    budget = 32 << 30
    try:
        dev = next(pipeline.unet.parameters()).device
        if dev.type == 'cuda':
            free, _ = torch.cuda.mem_get_info(dev)
            budget = min(budget, max(free // 4, 1 << 30))
    except Exception:                                   # no parameters / no device yet: keep the default
        pass
    return budget


class DiffusionHeatMapHooker(AggregateHooker):
    def __init__(self, pipeline, low_memory: bool = False, load_heads: bool = False, save_heads: bool = False,
                 data_dir: Optional[str] = None, *, accumulate: str = 'exact', tap: str = 'qk',
                 defer_steps: Optional[int] = None):
        """Positional arguments as in the reference (trace.py:23-30).  Keyword-only extras:
        ``accumulate`` = ``'exact'`` (running sums in the pipeline dtype, like the reference) or
        ``'float32'``; ``tap`` = ``'qk'`` (fused, default) or ``'probs'`` (materialised
        probabilities, bit-identical adds); ``defer_steps`` = denoising steps tapped per launch
        (0 = one launch per layer call; default ``$DAAM_DEFER_STEPS`` or 64, the most one launch takes).
        The Q / K of the recorded steps are kept alive until their launch: at most ``$DAAM_DEFER_BYTES``
        (default 32 GiB of the 288 GB; a 50-step SDXL-1024 generation holds 19.4 GB -- both CFG halves of
        every Q -- and runs as ONE tap launch, issued when the maps are first read)."""
        if tap not in ('qk', 'probs'):
            raise ValueError("tap must be 'qk' or 'probs'")
        h = pipeline.unet.config.sample_size * pipeline.vae_scale_factor
        self.latent_hw = 4096 if h == 512 or h == 1024 else 9216          # trace.py:32-33
        locate_middle = load_heads or save_heads
        self.locator = UNetCrossAttentionLocator(restrict={0} if low_memory else None,
                                                 locate_middle_block=locate_middle)
        modules_found = self.locator.locate(pipeline.unet)
        self.engine = HeatMapEngine(max(1, len(modules_found)), tokens=77, out_side=int(math.sqrt(self.latent_hw)),
                                    accumulate=accumulate,
                                    defer_steps=_default_defer() if defer_steps is None else defer_steps,
                                    defer_bytes=_default_defer_bytes(pipeline))
        self.all_heat_maps = RawHeatMapCollection(self.engine)
        self.last_prompt: str = ''
        self.last_image = None
        self.time_idx = 0
        self._gen_idx = 0
        self.tap_mode = tap

        hookers: List[ObjectHooker] = [
            UNetCrossAttentionHooker(m, self, layer_idx=idx, latent_hw=self.latent_hw, load_heads=load_heads,
                                     save_heads=save_heads, data_dir=data_dir)
            for idx, m in enumerate(modules_found)
        ]
        hookers.append(PipelineHooker(pipeline, self))
        if type(pipeline).__name__ == 'StableDiffusionXLPipeline':           # trace.py:55-56
            hookers.append(ImageProcessorHooker(pipeline.image_processor, self))
        super().__init__(hookers)
        self.pipe = pipeline

    def time_callback(self, *args, **kwargs):
        self.time_idx += 1

    @property
    def layer_names(self):
        return self.locator.layer_names

    def _unhook_impl(self):
        super()._unhook_impl()
        self.engine.flush()

    def to_experiment(self, path, seed=None, id='.', subtype='.', **compute_kwargs):
        """trace.py:68-81."""
        from .experiment import GenerationExperiment
        return GenerationExperiment(self.last_image, self.compute_global_heat_map(**compute_kwargs).heat_maps,
                                    self.last_prompt, seed=seed, id=id, subtype=subtype, path=path,
                                    tokenizer=self.pipe.tokenizer)

    def compute_global_heat_map(self, prompt=None, factors=None, head_idx=None, layer_idx=None, normalize=False):
        """Aggregate over time (already summed by the tap), layers and heads (trace.py:83-132):
        per selected ``(factor, layer, head)`` key bicubic-resize the summed map to ``x*x``, clamp
        at 0, average the keys, keep ``len(tokenize(prompt)) + 2`` rows, optionally normalise per
        pixel over the content tokens.  Returns a ``GlobalHeatMap`` whose ``heat_maps`` is an
        fp32 device tensor."""
        if prompt is None:
            prompt = self.last_prompt
        try:
            maps = self.engine.global_heat_map(factors=factors, head_idx=head_idx, layer_idx=layer_idx)
        except LookupError:
            if head_idx is not None or layer_idx is not None:
                raise RuntimeError('No heat maps found for the given parameters.') from None
            raise RuntimeError('No heat maps found. Did you forget to call `with trace(...)` during generation?') \
                from None
        maps = maps[:len(self.pipe.tokenizer.tokenize(prompt)) + 2]            # 1 for SOS and 1 for padding
        if normalize:
            maps = self.engine.normalize_(maps)
        return GlobalHeatMap(self.pipe.tokenizer, prompt, maps)


class ImageProcessorHooker(ObjectHooker):
    """trace.py:135-147 (SDXL: remember the first post-processed image)."""

    def __init__(self, processor, parent_trace: 'trace'):
        super().__init__(processor)
        self.parent_trace = parent_trace

    def _hooked_postprocess(hk_self, _processor, *args, **kwargs):
        images = hk_self.monkey_super('postprocess', *args, **kwargs)
        hk_self.parent_trace.last_image = images[0]
        return images

    def _hook_impl(self):
        self.monkey_patch('postprocess', self._hooked_postprocess)


class PipelineHooker(ObjectHooker):
    """trace.py:150-186: single-prompt guard, reset of the running sums at every ``pipe()``
    call, bookkeeping of the last prompt / image."""

    def __init__(self, pipeline, parent_trace: 'trace'):
        super().__init__(pipeline)
        self.heat_maps = parent_trace.all_heat_maps
        self.parent_trace = parent_trace

    def _hooked_run_safety_checker(hk_self, pipe, image, *args, **kwargs):
        image, has_nsfw = hk_self.monkey_super('run_safety_checker', image, *args, **kwargs)
        if getattr(pipe, 'image_processor', None):
            if torch.is_tensor(image):
                images = pipe.image_processor.postprocess(image, output_type='pil')
            else:
                images = pipe.image_processor.numpy_to_pil(image)
        else:
            images = pipe.numpy_to_pil(image)
        hk_self.parent_trace.last_image = images[len(images) - 1]
        return image, has_nsfw

    def _hooked_check_inputs(hk_self, _pipe, prompt: Union[str, List[str]], *args, **kwargs):
        if not isinstance(prompt, str) and len(prompt) > 1:
            raise ValueError('Only single prompt generation is supported for heat map computation.')
        last_prompt = prompt if isinstance(prompt, str) else prompt[0]
        hk_self.heat_maps.clear()
        hk_self.parent_trace.last_prompt = last_prompt
        return hk_self.monkey_super('check_inputs', prompt, *args, **kwargs)

    def _hook_impl(self):
        self.monkey_patch('run_safety_checker', self._hooked_run_safety_checker, strict=False)  # absent in SDXL
        self.monkey_patch('check_inputs', self._hooked_check_inputs)


class UNetCrossAttentionHooker(ObjectHooker):
    """The attention processor installed on every located ``attn2`` (diffusers
    attention-processor protocol, reference trace.py:252-311)."""

    def __init__(self, module, parent_trace: 'trace', context_size: int = 77, layer_idx: int = 0,
                 latent_hw: int = 9216, load_heads: bool = False, save_heads: bool = False,
                 data_dir: Union[str, Path, None] = None):
        super().__init__(module)
        self.heat_maps = parent_trace.all_heat_maps
        self.context_size = context_size
        self.layer_idx = layer_idx
        self.latent_hw = latent_hw
        self.load_heads = load_heads
        self.save_heads = save_heads
        self.trace = parent_trace
        self.data_dir = Path(data_dir) if data_dir is not None else cache_dir() / 'heads'
        self.data_dir.mkdir(parents=True, exist_ok=True)                       # trace.py:217

    def _save_attn(self, attn_slice: torch.Tensor):
        torch.save(attn_slice, self.data_dir / f'{self.trace._gen_idx}.pt')

    def _load_attn(self) -> torch.Tensor:
        return torch.load(self.data_dir / f'{self.trace._gen_idx}.pt')

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **_ignored):
        batch_size, sequence_length, _ = hidden_states.shape
        attention_mask = attn.prepare_attention_mask(attention_mask, sequence_length, batch_size)
        query = attn.to_q(hidden_states)
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states
        elif attn.norm_cross is not None:
            encoder_hidden_states = attn.norm_cross(encoder_hidden_states)
        key = attn.to_k(encoder_hidden_states)
        value = attn.to_v(encoder_hidden_states)

        engine = self.trace.engine
        factor = int(math.sqrt(self.latent_hw // sequence_length))             # trace.py:285
        tapped = key.shape[1] == self.context_size and factor != 8             # trace.py:289
        fused = (self.trace.tap_mode == 'qk' and attention_mask is None and not self.save_heads
                 and not self.load_heads and not getattr(attn, 'upcast_softmax', False))

        if fused:
            self.trace._gen_idx += 1
            if tapped:
                engine.tap_qk(self.layer_idx, query, key, attn.heads, attn.scale, factor,
                              round_logits=not getattr(attn, 'upcast_attention', False))
            heads = attn.heads
            d = query.shape[-1] // heads
            q4 = query.view(batch_size, -1, heads, d).transpose(1, 2)
            k4 = key.view(batch_size, -1, heads, d).transpose(1, 2)
            v4 = value.view(batch_size, -1, heads, d).transpose(1, 2)
            out = F.scaled_dot_product_attention(q4, k4, v4, scale=attn.scale)
            hidden_states = out.transpose(1, 2).reshape(batch_size, -1, heads * d)
        else:
            query = attn.head_to_batch_dim(query)
            key = attn.head_to_batch_dim(key)
            value = attn.head_to_batch_dim(value)
            attention_probs = attn.get_attention_scores(query, key, attention_mask)
            if self.save_heads:
                self._save_attn(attention_probs)
            elif self.load_heads:
                attention_probs = self._load_attn()
            factor = int(math.sqrt(self.latent_hw // attention_probs.shape[1]))
            self.trace._gen_idx += 1
            if attention_probs.shape[-1] == self.context_size and factor != 8:
                engine.tap_probs(self.layer_idx, attention_probs, factor)
            hidden_states = torch.bmm(attention_probs, value)
            hidden_states = attn.batch_to_head_dim(hidden_states)

        hidden_states = attn.to_out[0](hidden_states)      # linear proj
        hidden_states = attn.to_out[1](hidden_states)      # dropout
        return hidden_states

    def _hook_impl(self):
        self.original_processor = self.module.processor
        self.module.set_processor(self)

    def _unhook_impl(self):
        self.module.set_processor(self.original_processor)

    @property
    def num_heat_maps(self):
        return len(self.heat_maps)


trace: Type[DiffusionHeatMapHooker] = DiffusionHeatMapHooker

"""``trace`` / ``DiffusionHeatMapHooker``: the reference's tracing API (``daam/trace.py``)
over the MI355X-native extraction path.

Differences from the reference are confined to *how* the per-layer work is done:
  * the cross-attention processor does not materialise ``attention_probs`` on the default
    path: the attention itself runs on ``daam_attend`` (softmax(QK^T)V with the reference's
    rounding points, one MFMA kernel; the framework's fused SDPA where that kernel does not
    apply), and the heat-map tap either happens inside the same kernel (``defer_steps=0``) or
    recomputes the conditional-half probabilities from the projected Q / K in ONE launch for
    several denoising steps of all layers (``daam_tap_qk_enqueue`` / ``daam_tap_flush``, the default);
  * ``compute_global_heat_map`` is one fused bicubic + clamp + mean kernel.
The materialised path (``get_attention_scores`` -> ``daam_tap_probs`` -> ``bmm``) is kept for
``save_heads`` / ``load_heads``, attention masks, and ``tap='probs'``.
"""
from __future__ import annotations

import math
import os
import weakref
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
    """Bytes of recorded Q / K a trace may keep alive between tap launches: ``$DAAM_DEFER_BYTES``, else 40 % of the
    device memory that is free when the trace is set up (never less than 1 GiB, never more than 128 GiB; 32 GiB when the device
    cannot be asked).
    An MI355X has 288 GB: an SDXL-1024 generation holds 19.4 GB for its one launch, and SDXL at 2048 x 2048 (1.55 GB per
    denoising step, both CFG halves of every Q) gets the 64 steps a launch can take -- 100 steps = 2 launches, each
    re-reading the 0.88 GB of running sums once, instead of the 5 a fixed 32 GiB forced."""
    env = os.environ.get('DAAM_DEFER_BYTES')
    if env:
        return int(env)
    budget = 32 << 30
    try:
        dev = next(pipeline.unet.parameters()).device
        if dev.type != 'cuda' and torch.cuda.is_available():
            # cpu-offloaded pipelines keep their parameters on the host and run on the current device
            dev = torch.device('cuda', torch.cuda.current_device())
        if dev.type == 'cuda':
            free, _ = torch.cuda.mem_get_info(dev)
            # what torch's caching allocator holds but has not handed out is as good as free for the Q / K to come
            free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
            # absolute ceiling 128 GiB: enough for the 64 steps one launch takes at SDXL-2048 (99 GB), and it bounds what several
            # traces that each see the same free memory (ranks sharing a device) can pin between them
            budget = min(max(int(free * 0.4), 1 << 30), 128 << 30)
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
        (default: 40 % of the device memory free at set-up; a 50-step SDXL-1024 generation holds 19.4 GB -- both CFG
        halves of every Q -- and runs as ONE tap launch, issued when the maps are first read)."""
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
                                    defer_bytes=_default_defer_bytes(pipeline), reuse_context=True)
        self.all_heat_maps = RawHeatMapCollection(self.engine)
        self.last_prompt: str = ''
        self.last_image = None
        self.time_idx = 0
        self._gen_idx = 0
        self.tap_mode = tap

        # the child hookers see this object through a weak proxy: no parent <-> child reference cycle, so a trace
        # that goes out of scope releases its context and running sums immediately
        me = weakref.proxy(self)
        hookers: List[ObjectHooker] = [
            UNetCrossAttentionHooker(m, me, layer_idx=idx, latent_hw=self.latent_hw, load_heads=load_heads,
                                     save_heads=save_heads, data_dir=data_dir)
            for idx, m in enumerate(modules_found)
        ]
        hookers.append(PipelineHooker(pipeline, me))
        if type(pipeline).__name__ == 'StableDiffusionXLPipeline':           # trace.py:55-56
            hookers.append(ImageProcessorHooker(pipeline.image_processor, me))
        super().__init__(hookers)
        self.pipe = pipeline

    def time_callback(self, *args, **kwargs):
        self.time_idx += 1

    @property
    def layer_names(self):
        return self.locator.layer_names

    def _hook_impl(self):
        super()._hook_impl()
        # the installed processors / patched pipeline methods see the trace through weak proxies (no reference cycle
        # while idle); WHILE hooked they must keep it alive -- ``trace(pipe).hook()`` without holding on to the object is
        # legal with the reference -- so the hookers pin it until unhook() breaks the cycle again
        for member in self.module:
            member._pinned_trace = self

    def _unhook_impl(self):
        super()._unhook_impl()
        for member in self.module:
            member._pinned_trace = None
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
        n_rows = len(self.pipe.tokenizer.tokenize(prompt)) + 2                 # 1 for SOS and 1 for padding (trace.py:127)
        try:
            # the crop is handed to the finalize: rows nobody reads are not computed (daam_finalize n_rows, ABI v6)
            maps = self.engine.global_heat_map(factors=factors, head_idx=head_idx, layer_idx=layer_idx, n_rows=n_rows)
        except LookupError:
            if head_idx is not None or layer_idx is not None:
                raise RuntimeError('No heat maps found for the given parameters.') from None
            raise RuntimeError('No heat maps found. Did you forget to call `with trace(...)` during generation?') \
                from None
        maps = maps[:n_rows]
        if normalize:
            maps = self.engine.normalize_(maps)
        return GlobalHeatMap(self.pipe.tokenizer, prompt, maps)


class _CallInterceptor(ObjectHooker):
    """Wraps methods of one pipeline-side object.  ``WRAPS`` lists ``(attribute, handler name, strict)``;
    a handler receives the patched object first and reaches the original through ``self.monkey_super``."""

    WRAPS = ()

    def __init__(self, target, parent_trace: 'trace'):
        super().__init__(target)
        self.parent_trace = parent_trace

    def _hook_impl(self):
        for attribute, handler, strict in self.WRAPS:
            self.monkey_patch(attribute, getattr(self, handler), strict=strict)


class ImageProcessorHooker(_CallInterceptor):
    """SDXL pipelines post-process through ``pipe.image_processor``: remember the first image it returns
    (reference trace.py:135-147)."""

    WRAPS = (('postprocess', '_after_postprocess', True),)

    def _after_postprocess(self, _processor, *args, **kwargs):
        images = self.monkey_super('postprocess', *args, **kwargs)
        self.parent_trace.last_image = images[0]
        return images


class PipelineHooker(_CallInterceptor):
    """What a ``pipe(prompt)`` call means for the trace (reference trace.py:150-186): exactly one prompt, the
    running sums start from zero, the prompt and the generated image are remembered."""

    WRAPS = (('run_safety_checker', '_after_safety_checker', False),      # absent in SDXL pipelines
             ('check_inputs', '_before_generation', True))

    def __init__(self, pipeline, parent_trace: 'trace'):
        super().__init__(pipeline, parent_trace)
        self.heat_maps = parent_trace.all_heat_maps

    def _before_generation(self, _pipe, prompt: Union[str, List[str]], *args, **kwargs):
        single = isinstance(prompt, str)
        if not single and len(prompt) > 1:
            raise ValueError('Only single prompt generation is supported for heat map computation.')
        self.heat_maps.clear()                                             # RawHeatMapCollection.clear -> daam_reset
        self.parent_trace.last_prompt = prompt if single else prompt[0]
        return self.monkey_super('check_inputs', prompt, *args, **kwargs)

    def _after_safety_checker(self, pipe, image, *args, **kwargs):
        checked = self.monkey_super('run_safety_checker', image, *args, **kwargs)
        processor = getattr(pipe, 'image_processor', None)
        if not processor:
            pils = pipe.numpy_to_pil(checked[0])
        elif torch.is_tensor(checked[0]):
            pils = processor.postprocess(checked[0], output_type='pil')
        else:
            pils = processor.numpy_to_pil(checked[0])
        self.parent_trace.last_image = pils[-1]
        return checked


class UNetCrossAttentionHooker(ObjectHooker):
    """The attention processor installed on every located ``attn2`` (diffusers attention-processor protocol;
    replaces the reference's processor, trace.py:252-311).

    Default route: the model's output comes from ``HeatMapEngine.attend`` (``daam_attend``; the framework's fused SDPA
    for calls that kernel does not take) and the heat-map tap gets the projected Q / K (fused into the same kernel on
    an immediate trace, ``HeatMapEngine.tap_qk`` otherwise -- nothing ``[BH, hw, 77]``-sized is ever materialised).  Materialised route
    (``get_attention_scores`` -> ``tap_probs`` -> ``bmm``) for attention masks, ``upcast_softmax``,
    ``save_heads`` / ``load_heads`` and ``tap='probs'``."""

    def __init__(self, module, parent_trace: 'trace', context_size: int = 77, layer_idx: int = 0,
                 latent_hw: int = 9216, load_heads: bool = False, save_heads: bool = False,
                 data_dir: Union[str, Path, None] = None):
        super().__init__(module)
        self.trace = parent_trace
        self.heat_maps = parent_trace.all_heat_maps
        self.layer_idx = layer_idx
        self.context_size = context_size
        self.latent_hw = latent_hw
        self.save_heads, self.load_heads = save_heads, load_heads
        self.data_dir = cache_dir() / 'heads' if data_dir is None else Path(data_dir)
        self.data_dir.mkdir(parents=True, exist_ok=True)                   # the reference creates it eagerly too (:217)
        self._factors = {}                                                 # positions -> factor

    # -- installation ----------------------------------------------------------------------------
    def _hook_impl(self):
        attn = self.module
        self.original_processor = attn.processor
        # per-generation constants of the default route
        self._heads, self._scale = attn.heads, attn.scale
        self._round_logits = not getattr(attn, 'upcast_attention', False)
        self._fusable = (self.trace.tap_mode == 'qk' and not self.save_heads and not self.load_heads
                         and not getattr(attn, 'upcast_softmax', False))
        self._tap_qk = self.trace.engine.tap_qk           # the C++ recorder's entry point on a deferred trace
        # attention itself on the library's kernel (fp16, 77 keys, head_dim a multiple of 8 up to 160), with the tap fused in on an immediate
        # trace; DAAM_NO_ATTEND=1 keeps the framework's fused SDPA for the model's output
        self._attend = None if os.environ.get('DAAM_NO_ATTEND') else self.trace.engine.attend
        attn.set_processor(self)

    def _unhook_impl(self):
        self.module.set_processor(self.original_processor)

    @property
    def num_heat_maps(self):
        return len(self.heat_maps)

    # -- heads cache (one file per processor call of the generation, reference :246-250) --------------
    def _heads_file(self) -> Path:
        return self.data_dir / f'{self.trace._gen_idx}.pt'

    def _save_attn(self, attn_slice: torch.Tensor):
        torch.save(attn_slice, self._heads_file())

    def _load_attn(self) -> torch.Tensor:
        return torch.load(self._heads_file())

    # -- the processor call ------------------------------------------------------------------------
    def _factor(self, positions: int) -> int:
        factor = self._factors.get(positions)
        if factor is None:
            factor = self._factors[positions] = int(math.sqrt(self.latent_hw // positions))   # trace.py:285
        return factor

    def _is_tapped(self, tokens: int, factor: int) -> bool:
        return tokens == self.context_size and factor != 8                 # trace.py:289

    def _fused(self, attn, hidden_states, context):
        """Default route.  Runs 60-70 times per denoising step: the attribute lookups that cannot change during
        a generation (heads, scale, the engine's recorder entry point) are resolved once in ``_hook_impl``."""
        query, key, value = attn.to_q(hidden_states), attn.to_k(context), attn.to_v(context)
        self.trace._gen_idx += 1
        batch, positions, channels = query.shape
        factor = self._factor(positions)
        tapped = factor != 8 and key.shape[1] == self.context_size          # trace.py:289
        if self._attend is not None and key.shape[1] == self.context_size:
            out = self._attend(self.layer_idx, query, key, value, self._heads, self._scale, factor, self._round_logits, tapped)
            if out is not None:
                return out
        if tapped:
            self._tap_qk(self.layer_idx, query, key, self._heads, self._scale, factor, self._round_logits)
        heads = self._heads
        head_dim = channels // heads
        out = F.scaled_dot_product_attention(query.view(batch, -1, heads, head_dim).transpose(1, 2),
                                             key.view(batch, -1, heads, head_dim).transpose(1, 2),
                                             value.view(batch, -1, heads, head_dim).transpose(1, 2), scale=self._scale)
        return out.transpose(1, 2).reshape(batch, -1, channels)

    def _materialised(self, attn, hidden_states, context, attention_mask):
        query, key, value = (attn.head_to_batch_dim(t)
                             for t in (attn.to_q(hidden_states), attn.to_k(context), attn.to_v(context)))
        probs = attn.get_attention_scores(query, key, attention_mask)
        if self.save_heads:
            self._save_attn(probs)
        elif self.load_heads:
            probs = self._load_attn()
        self.trace._gen_idx += 1
        factor = self._factor(probs.shape[1])
        if self._is_tapped(probs.shape[-1], factor):
            self.trace.engine.tap_probs(self.layer_idx, probs, factor)
        return attn.batch_to_head_dim(torch.bmm(probs, value))

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, **_ignored):
        if encoder_hidden_states is None:
            context = hidden_states
        elif attn.norm_cross is None:
            context = encoder_hidden_states
        else:
            context = attn.norm_cross(encoder_hidden_states)
        if self._fusable and attention_mask is None:       # prepare_attention_mask(None, ...) is None: nothing to prepare
            mixed = self._fused(attn, hidden_states, context)
        else:
            batch, positions, _ = hidden_states.shape
            attention_mask = attn.prepare_attention_mask(attention_mask, positions, batch)
            mixed = self._materialised(attn, hidden_states, context, attention_mask)
        return attn.to_out[1](attn.to_out[0](mixed))                       # output projection, dropout


trace: Type[DiffusionHeatMapHooker] = DiffusionHeatMapHooker

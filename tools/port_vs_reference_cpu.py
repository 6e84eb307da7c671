#!/usr/bin/env python
"""Times the UNMODIFIED reference's hook path next to the torch port (oracle/torch_hooks.py) that bench.py times as
``cpu_baseline`` on the GPU box -- same inputs, same host, same thread counts.  Runs only where ``/root/reference`` exists
(the build container); the committed result is profiles/r03_port_vs_reference_cpu.json.

    python tools/port_vs_reference_cpu.py [sdxl|sd15] > profiles/r03_port_vs_reference_cpu.json

Reference side (imported through oracle/fake_diffusers.py's stubs, executed as it is): per denoising step and hooked layer
``UNetCrossAttentionHooker._unravel_attn`` (daam/trace.py:219-244) + ``RawHeatMapCollection.update`` per head
(daam/heatmap.py:153-156) -- the DAAM-specific part of ``__call__`` (trace.py:285-294) -- and one
``DiffusionHeatMapHooker.compute_global_heat_map`` (trace.py:83-132) over all keys.  Port side: ``th.tap`` / ``th.global_heat_map``.
Inputs: fp32 probabilities [2H, hw, 77] of the real layer shapes (SURVEY.md section 8 topology), one denoising step + finalize.
"""
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fake_diffusers as fd       # noqa: E402
from oracle import torch_hooks as th          # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else 'sdxl'
    daam, trace_mod = fd.import_reference()
    layers = th.execution_order(th.topology(kind))
    cache = {}
    for (_, heads, side, _) in layers:
        if (heads, side) not in cache:
            cache[(heads, side)] = torch.rand(2 * heads, side * side, 77)
    latent_hw = 4096

    # the reference's objects: a hooker per layer (its _unravel_attn is a method, its heat_maps the shared collection) on a
    # stand-in trace object that carries only what the two functions read
    class _Parent:
        def __init__(self):
            self.all_heat_maps = daam.heatmap.RawHeatMapCollection()
            self.latent_hw = latent_hw
            self.last_prompt = ' '.join(['w'] * 75)

    def ref_step(parent, hookers):
        for (layer, heads, side, _d), hk in zip(layers, hookers):
            probs = cache[(heads, side)]
            factor = int(math.sqrt(latent_hw // probs.shape[1]))                       # trace.py:285
            if probs.shape[-1] == 77 and factor != 8:                                  # trace.py:289
                maps = hk._unravel_attn(probs)                                         # trace.py:292
                for head_idx, heatmap in enumerate(maps):                              # trace.py:293-294
                    parent.all_heat_maps.update(factor, layer, head_idx, heatmap)

    def make_ref():
        parent = _Parent()
        hookers = []
        for (layer, heads, side, _d) in layers:
            hk = object.__new__(trace_mod.UNetCrossAttentionHooker)                    # no module to hook: the methods are what is timed
            hk.context_size, hk.layer_idx, hk.latent_hw = 77, layer, latent_hw
            hookers.append(hk)
        return parent, hookers

    def ref_global(parent):
        tr = object.__new__(trace_mod.DiffusionHeatMapHooker)
        tr.all_heat_maps, tr.latent_hw, tr.last_prompt = parent.all_heat_maps, latent_hw, parent.last_prompt
        tr.pipe = type('P', (), {'tokenizer': fd.FakeTokenizer()})()
        return tr.compute_global_heat_map()

    out = dict(kind=kind, layers=len(layers), cpu=open('/proc/cpuinfo').read().split('model name')[1].split('\n')[0].strip(': \t'),
               torch=torch.__version__, reference='castorini/daam v0.2.0, unmodified, imported through oracle/fake_diffusers.py stubs',
               sample='1 denoising step over every hooked layer (fp32 probabilities of the real shapes) + 1 compute_global_heat_map',
               by_threads={})
    for threads in sorted({1, os.cpu_count() or 1}):
        torch.set_num_threads(threads)
        rec = {}
        for name in ('reference', 'port'):
            best_step, best_fin = float('inf'), float('inf')
            for _ in range(2):
                if name == 'reference':
                    parent, hookers = make_ref()
                    t0 = time.perf_counter()
                    ref_step(parent, hookers)
                    t1 = time.perf_counter()
                    g = ref_global(parent).heat_maps
                    t2 = time.perf_counter()
                else:
                    raw = th.RawMaps()
                    t0 = time.perf_counter()
                    for (layer, heads, side, _d) in layers:
                        th.tap(raw, layer, cache[(heads, side)], latent_hw)
                    t1 = time.perf_counter()
                    g2 = th.global_heat_map(raw, latent_hw)
                    t2 = time.perf_counter()
                best_step, best_fin = min(best_step, t1 - t0), min(best_fin, t2 - t1)
            rec[name] = dict(ms_per_denoise_step=round(best_step * 1e3, 2), finalize_s=round(best_fin, 4))
        rec['port_over_reference'] = dict(step=round(rec['port']['ms_per_denoise_step'] / rec['reference']['ms_per_denoise_step'], 3),
                                          finalize=round(rec['port']['finalize_s'] / rec['reference']['finalize_s'], 3))
        out['by_threads'][str(threads)] = rec
    out['same_result'] = bool(torch.allclose(g, g2[:g.shape[0]], rtol=0, atol=1e-6))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()

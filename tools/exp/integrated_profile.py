"""Debug: where the integrated-harness overhead of tests/test_gpu_integration.py comes from (MI355X)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import torch
from oracle import fake_diffusers as fd
import test_gpu_integration as T
import daam_amd

pipe = fd.make_pipe('sdxl', device='cuda:0', dtype=torch.float16, batch=2, seed=3, mini=False, identity_proj=False)
T._resident_inputs(pipe, 4)
mods = [s.module for s in pipe.unet.execution_order()]
for m in mods:
    m.set_processor(T._SdpaProcessor())
prompt, steps = 'a photo of a monkey', 20


def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps / steps * 1e3


plain = timed(lambda: pipe(prompt, num_inference_steps=steps))
print(f'plain SDPA processor            {plain:7.3f} ms/step')


def traced(noop_tap=False, finalize=True, **kw):
    def run():
        with daam_amd.trace(pipe, **kw) as tc:
            if noop_tap:
                for h in tc.module[:-2]:
                    h._tap_qk = lambda *a: None
            pipe(prompt, num_inference_steps=steps)
            if finalize and not noop_tap:
                tc.compute_global_heat_map()
    return run


for name, fn in [('trace, tap replaced by a no-op', traced(noop_tap=True)),
                 ('trace, no compute_global_heat_map', traced(finalize=False)),
                 ('trace (default)', traced()),
                 ('trace, defer_steps=4', traced(defer_steps=4)),
                 ('trace, defer_steps=0 (immediate)', traced(defer_steps=0))]:
    t = timed(fn)
    print(f'{name:38s} {t:7.3f} ms/step  (+{t - plain:.3f})')

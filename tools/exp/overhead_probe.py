"""Where the integrated extraction overhead goes (bench.py's integrated leg, taken apart).

For the plain and the traced arm of the synthetic SDXL-1024 stack: host time to ISSUE a generation (python returns from
pipe(), no sync), GPU time of the stack (events around pipe()), and for the traced arm the time of flush + finalize.
Paired, interleaved generations; medians.

    python tools/exp/overhead_probe.py [steps] [reps]
"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    import daam_amd
    from tools.synthetic_unet import SyntheticPipeline
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 9
    pipe = SyntheticPipeline('sdxl', 128, device='cuda:0')
    prompt = 'a photo of a monkey riding a bicycle'

    def ev():
        return torch.cuda.Event(enable_timing=True)

    def plain():
        torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        t0 = time.perf_counter()
        e0.record()
        pipe(prompt, num_inference_steps=steps)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        return dict(issue=t1 - t0, wall=t2 - t0, gpu_stack=e0.elapsed_time(e1) * 1e-3)

    def traced():
        torch.cuda.synchronize()
        e0, e1, e2 = ev(), ev(), ev()
        t0 = time.perf_counter()
        with daam_amd.trace(pipe) as tc:
            th = time.perf_counter()
            e0.record()
            pipe(prompt, num_inference_steps=steps)
            e1.record()
            t1 = time.perf_counter()
            maps = tc.compute_global_heat_map().heat_maps
            e2.record()
            t2 = time.perf_counter()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        del maps
        return dict(hook=th - t0, issue=t1 - th, ghm_host=t2 - t1, unhook=t3 - t2, wall=t4 - t0,
                    gpu_stack=e0.elapsed_time(e1) * 1e-3, gpu_ghm=e1.elapsed_time(e2) * 1e-3)

    # phase timers inside compute_global_heat_map (host seconds, accumulated per generation)
    from daam_amd import engine as eng_mod
    phases = {}

    def timed_method(cls, name):
        orig = getattr(cls, name)

        def wrapper(self, *a, **k):
            t = time.perf_counter()
            try:
                return orig(self, *a, **k)
            finally:
                phases[name] = phases.get(name, 0.0) + time.perf_counter() - t
        setattr(cls, name, wrapper)
    if os.environ.get('PROBE_PHASES', '1') == '1':
        for name in ('flush', '_drop_recorded', 'finalize', '_launch_stream'):
            if hasattr(eng_mod.HeatMapEngine, name):
                timed_method(eng_mod.HeatMapEngine, name)

    for _ in range(2):
        plain()
        traced()
    phases.clear()
    P, T = [], []
    for _ in range(reps):
        P.append(plain())
        T.append(traced())

    def med(rows, k):
        return statistics.median(r[k] for r in rows)
    out = dict(steps=steps, reps=reps,
               plain={k: round(med(P, k) / steps * 1e3, 4) for k in P[0]},
               traced={k: round(med(T, k) / steps * 1e3, 4) for k in T[0]},
               paired_overhead_ms_per_step=round(statistics.median(t['wall'] - p['wall'] for p, t in zip(P, T)) / steps * 1e3, 4),
               paired_overheads=[round((t['wall'] - p['wall']) / steps * 1e3, 4) for p, t in zip(P, T)],
               phases_ms_per_generation={k: round(v / reps * 1e3, 3) for k, v in phases.items()},
               unit='ms per denoising step')
    print(json.dumps(out))


if __name__ == '__main__':
    main()

"""Soak test of the deferred tap launches: the same generation (same resident Q / K) N times back to back; after every one the running sums of
every layer must equal the first generation's BIT FOR BIT (the tap has no atomics and no order freedom: any difference is a race in a
kernel's LDS / barrier protocol or in the launch machinery).  Checked on the device (torch.equal per layer), one host sync per generation.

    python tools/exp/soak.py [sd15|sdxl1024|sdxl1024_bf16] [seconds]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from daam_amd.engine import HeatMapEngine  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'sd15'
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    wl = bench.WORKLOADS[name]
    layers = bench.topology(wl['kind'], wl['latent'])
    steps = 50
    sets = bench.make_inputs(layers, steps, dev, seed=99, dtype=getattr(torch, wl.get('dtype', 'float16')))
    calls = bench.call_lists(layers, sets, 64)
    eng = HeatMapEngine(len(layers), tokens=77, out_side=64, accumulate=wl.get('accumulate', 'exact'), defer_steps=64,
                        defer_bytes=bench.default_defer_bytes(dev))

    def generation():
        eng.clear()
        for t in range(steps):
            for a in calls[t]:
                eng.tap_qk(*a)
        eng.flush()
        return [v for _, v in eng.items()]
    first = [v.clone() for v in generation()]
    torch.cuda.synchronize()
    assert sum(float(v.float().abs().sum()) for v in first) > 0
    n, bad, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        cur = generation()
        same = all(torch.equal(a, b) for a, b in zip(cur, first))
        n += 1
        if not same:
            bad += 1
            worst = max(float((a.float() - b.float()).abs().max()) for a, b in zip(cur, first))
            print(f'generation {n}: sums differ from the first generation (max-abs {worst})', file=sys.stderr, flush=True)
    out = dict(workload=name, generations=n, seconds=round(time.perf_counter() - t0, 1), differing_generations=bad, keys=len(first),
               flush=eng.last_flush())
    print(json.dumps(out))
    eng.close()
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

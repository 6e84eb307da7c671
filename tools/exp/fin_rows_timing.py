#!/usr/bin/env python
"""Round 6 (ABI v6): time of ``daam_finalize`` against ``n_rows`` -- the crop of daam/trace.py:127 applied before the work -- on the SDXL-1024
key set (1000 x2 keys + 100 same-size keys) for the three dtypes the sums can have.  HIP events around 40 back-to-back calls per point,
after a warm-up.  Writes gpurun_out/fin_rows_timing.json.

    python tools/exp/fin_rows_timing.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from daam_amd.engine import HeatMapEngine


def main():
    dev = torch.device('cuda', 0)
    layers = bench.topology('sdxl', 128)
    out = {}
    for name, dtype, accumulate in (('f16', torch.float16, 'exact'), ('bf16', torch.bfloat16, 'exact'), ('f32', torch.float16, 'float32')):
        sets = bench.make_inputs(layers, 2, dev, 1, dtype=dtype)
        eng = HeatMapEngine(len(layers), tokens=77, out_side=64, accumulate=accumulate, defer_steps=4)
        for t in range(4):
            for (layer, heads, side, d), (q, k) in zip(layers, sets[t % 2]):
                eng.tap_qk(layer, q, k, heads, d ** -0.5, 64 // side if side <= 64 else 0)
        eng.flush()
        rows = {}
        for n_rows in (77, 6, 12, 24, 40, 60, 77):
            for _ in range(10):
                eng.global_heat_map(n_rows=n_rows)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(40):
                eng.global_heat_map(n_rows=n_rows)
            b.record()
            torch.cuda.synchronize()
            rows[n_rows] = round(a.elapsed_time(b) / 40 * 1e3, 1)
        out[name] = dict(us_per_call=rows, kernels=eng.last_kernels(1))
        print(name, rows, eng.last_kernels(1), flush=True)
        eng.close()
        del sets
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/fin_rows_timing.json', 'w'), indent=1)


if __name__ == '__main__':
    main()

"""Where the SD-v1.5 tap launch (tap_slab_kernel, 1120 workgroups: 640-byte slabs x 32-pixel tiles) spends its time: per-workgroup stamps
of a -DDAAM_SLAB_TIMING build.

    python -c "from daam_amd import build; build.build_variant('tools/exp/libdaam_stime.so', ['-DDAAM_SLAB_TIMING'])"   (in the container)
    DAAM_HIP_LIB=tools/exp/libdaam_stime.so python tools/exp/slab_timeline.py [sd15|sdxl1024]                          (on the GPU box)

Per head_dim class: workgroups, when they start (first round / later), how long their step loop runs, the share of the loop's shader
cycles wave 0 spent in the DMA wait + barrier at the head of every sub-step, shader clock; and the launch's span."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from daam_amd import _native as nat  # noqa: E402
from daam_amd.engine import HeatMapEngine  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'sd15'
    lib = nat.load()
    if not hasattr(lib, 'daam_debug_dump_slab'):
        raise SystemExit('needs a -DDAAM_SLAB_TIMING build (DAAM_HIP_LIB)')
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    wl = bench.WORKLOADS[name]
    layers = bench.topology(wl['kind'], wl['latent'])
    steps = 50
    sets = bench.make_inputs(layers, 25 if name != 'sd15' else 50, dev, seed=1234)
    calls = bench.call_lists(layers, sets, 64)
    eng = HeatMapEngine(len(layers), tokens=77, out_side=64, accumulate='exact', defer_steps=64,
                        defer_bytes=bench.default_defer_bytes(dev))
    for _ in range(40):
        bench.one_generation(eng, calls, steps)
    torch.cuda.synchronize()
    buf = np.zeros((4096, 8), dtype=np.uint64)
    nat.check(lib.daam_debug_dump_slab(buf.ctypes.data_as(ctypes.c_void_p)))
    n = int((buf[:, 3] > 0).sum())
    b = buf[:n].astype(np.int64)
    t0 = b[:, 0].min()
    us = (b[:, :4] - t0) / 100.0                              # 100 MHz counter -> microseconds
    d = b[:, 6] & 0xffff
    xcc = (b[:, 7] >> 32) & 0xf
    cu = (b[:, 7] >> 8) & 0xf
    se = (b[:, 7] >> 13) & 0x7
    out = dict(workload=name, workgroups=n, span_us=round(float(us[:, 3].max()), 1),
               last_start_us=round(float(us[:, 0].max()), 1), classes=[])
    for hd in sorted(set(d.tolist())):
        m = d == hd
        start, loop, total = us[m, 0], us[m, 2] - us[m, 1], us[m, 3] - us[m, 0]
        first = start < 20.0
        rec = dict(head_dim=int(hd), workgroups=int(m.sum()), in_first_round=int(first.sum()),
                   loop_us_first_round=[round(float(np.percentile(loop[first], q)), 1) for q in (5, 50, 95)] if first.any() else None,
                   loop_us_later=[round(float(np.percentile(loop[~first], q)), 1) for q in (5, 50, 95)] if (~first).any() else None,
                   start_us_later=[round(float(np.percentile(start[~first], q)), 1) for q in (5, 50, 95)] if (~first).any() else None,
                   end_us=[round(float(np.percentile(us[m, 3], q)), 1) for q in (5, 50, 95, 100)],
                   prologue_us=round(float(np.median(us[m, 1] - us[m, 0])), 2), epilogue_us=round(float(np.median(us[m, 3] - us[m, 2])), 2),
                   wait_share_of_loop=round(float(np.median((b[m, 4] & 0xffffffff) / np.maximum(b[m, 5], 1))), 3),
                   wait3_share_of_loop=round(float(np.median((b[m, 4] >> 32) / np.maximum(b[m, 5], 1))), 3),
                   loop_cycles_per_step=round(float(np.median(b[m, 5] / 50.0)), 0),
                   mhz=round(float(np.median(b[m, 5] / np.maximum(loop, 1e-3))), 0))
        out['classes'].append(rec)
        print(rec, file=sys.stderr, flush=True)
    # how many workgroups are resident over time (10 us bins)
    edges = np.arange(0, us[:, 3].max() + 10, 10.0)
    out['resident_per_10us'] = [int(((us[:, 0] <= e) & (us[:, 3] > e)).sum()) for e in edges]
    out['per_xcd_end_us'] = [round(float(us[xcc == x, 3].max()), 1) for x in range(8) if (xcc == x).any()]
    print(dict(span_us=out['span_us'], last_start_us=out['last_start_us'], resident=out['resident_per_10us'], xcd_end=out['per_xcd_end_us']),
          file=sys.stderr, flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', f'slab_timeline_{name}.json'), 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

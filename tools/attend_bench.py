"""daam_attend on the SDXL-1024 layer set: the kernel with and without the fused tap, next to the attention the stock
processor runs (torch's fused SDPA) and to the stand-alone immediate tap.

    python tools/attend_bench.py [denoise steps] [reps] [attend|attend_fused_tap]     # one JSON line
    rocprofv3 --kernel-trace --stats ... -- python tools/attend_bench.py 10 2   # per-kernel durations of the same loops

Every loop issues, per denoising step, one call per hooked layer (60 for SDXL) on device-resident Q / K / V; times are
HIP-event times of the whole loop divided by the steps (they include launch gaps when the host is the bottleneck:
`issue_ms_per_step` says how long the host needed).  Algorithmic bytes of one attend call = Q + out (both CFG halves)
+ K + V, and with the tap + one read-modify-write of the layer's sums.
"""
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def measure(steps=50, reps=5, dev=None, only=None):
    import bench
    from daam_amd.engine import HeatMapEngine
    dev = torch.device('cuda:0') if dev is None else dev
    layers = bench.topology('sdxl', 128)
    sets = bench.make_inputs(layers, 2, dev, 0)
    g = torch.Generator(device=dev).manual_seed(1)
    vals = [torch.randn(2, 77, heads * d, generator=g, device=dev, dtype=torch.float16) for _, heads, _, d in layers]
    calls = bench.call_lists(layers, sets, 64)

    def loop(fn):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for s in range(steps):
            for i, a in enumerate(calls[s % len(calls)]):
                fn(i, a)
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / steps, (t1 - t0) * 1e3 / steps

    def best(fn, prepare=None):
        out = []
        for _ in range(reps + 1):
            if prepare:
                prepare()
            out.append(loop(fn))
        out = sorted(out[1:])
        return dict(ms_per_step=round(out[len(out) // 2][0], 4), issue_ms_per_step=round(out[len(out) // 2][1], 4))

    fused = HeatMapEngine(60, defer_steps=0)
    plain = HeatMapEngine(60, defer_steps=0)

    def sdpa(i, a):
        layer, q, k, heads, scale, factor = a
        b, _, c = q.shape
        d = c // heads
        v = vals[i]
        out = F.scaled_dot_product_attention(q.view(b, -1, heads, d).transpose(1, 2), k.view(b, -1, heads, d).transpose(1, 2),
                                             v.view(b, -1, heads, d).transpose(1, 2), scale=scale)
        return out.transpose(1, 2).reshape(b, -1, c)

    def attend_only(i, a):
        layer, q, k, heads, scale, factor = a
        return fused.attend(layer, q, k, vals[i], heads, scale, factor, True, False)

    def attend_tap(i, a):
        layer, q, k, heads, scale, factor = a
        return fused.attend(layer, q, k, vals[i], heads, scale, factor, True, True)

    def tap_only(i, a):
        plain.tap_qk(*a)

    res = dict(steps=steps, reps=reps, layers=len(layers))
    if only:                                             # one loop only (PMC passes: one kind of launch per kernel name)
        fn, prep = dict(attend=(attend_only, None), attend_fused_tap=(attend_tap, fused.clear))[only]
        res[only] = best(fn, prepare=prep)
        return res
    res['torch_sdpa'] = best(sdpa)
    res['attend'] = best(attend_only)
    res['attend_fused_tap'] = best(attend_tap, prepare=fused.clear)
    res['immediate_tap_alone'] = best(tap_only, prepare=plain.clear)
    qo = sum(2 * 2 * side * side * heads * d * 2 for _, heads, side, d in layers)          # Q + out, both CFG halves
    kv = sum(2 * 2 * 77 * heads * d * 2 for _, heads, side, d in layers)
    rmw = sum(2 * heads * 77 * side * side * 2 for _, heads, side, d in layers)              # fp16 sums, read + write
    res['algorithmic_bytes_per_step'] = dict(attend=qo + kv, attend_fused_tap=qo + kv + rmw)
    for name, b in (('attend', qo + kv), ('attend_fused_tap', qo + kv + rmw)):
        res[name]['gbps'] = round(b / (res[name]['ms_per_step'] * 1e-3) / 1e9, 1)
        res[name]['frac_of_hbm_peak'] = round(res[name]['gbps'] / 8000.0, 4)   # loop time incl. launch gaps: a lower bound for the kernel
    res['fused_tap_cost_ms_per_step'] = round(res['attend_fused_tap']['ms_per_step'] - res['attend']['ms_per_step'], 4)
    # parity of what the two paths left behind: the fused sums equal the stand-alone tap's, bit for bit
    a, b = dict(fused.items()), dict(plain.items())
    res['sums_bit_identical'] = bool(list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a))
    fused.close()
    plain.close()
    return res


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    only = sys.argv[3] if len(sys.argv) > 3 else None
    print(json.dumps(measure(steps, reps, only=only)))


if __name__ == '__main__':
    main()

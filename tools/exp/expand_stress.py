#!/usr/bin/env python
"""Round 6: hunt for the one-off mismatch of ``WordHeatMap.expand_as(absolute=True)`` seen once in a -n 4 GPU test run (64 of 16384
elements of the 128 x 128 output off by up to 0.023).  Repeats the calls of tests/test_gpu_parity.py's golden test on a fixed word map,
alternating the normalised and the absolute form (the same allocator blocks are handed out again and again), optionally while a
second process keeps the GPU busy, and reports every call whose result differs from the first one of its kind."""
import argparse, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=20000)
    ap.add_argument('--noise', type=int, default=0, help='run that many busy child processes beside the loop')
    ap.add_argument('--child', action='store_true')
    a = ap.parse_args()
    if a.child:
        x = torch.randn(4096, 4096, device='cuda', dtype=torch.float16)
        t0 = time.time()
        while time.time() - t0 < 120:
            y = x @ x
            _ = y[:64, :64].cpu()
        return
    kids = [subprocess.Popen([sys.executable, __file__, '--child']) for _ in range(a.noise)]
    from daam_amd.heatmap import WordHeatMap
    from daam_amd import engine as E
    g = torch.Generator(device='cuda').manual_seed(3)
    maps = torch.rand(12, 64, 64, generator=g, device='cuda') * 0.2
    word = E.word_heat_map(maps, [2, 3])
    whm = WordHeatMap(word, 'w')
    class Img: size = (128, 128)
    ref = {}
    bad = []
    t0 = time.time()
    for i in range(a.iters):
        # what the golden test does in between: global maps of changing row counts come and go in the same allocator pool
        junk = torch.empty(6 + i % 5, 64, 64, device='cuda').fill_(float(i))
        for absolute in (False, True):
            got = whm.expand_as(Img(), absolute=absolute).numpy()
            if absolute not in ref:
                ref[absolute] = got.copy()
            elif not np.array_equal(got, ref[absolute]):
                d = np.flatnonzero(got.ravel() != ref[absolute].ravel())
                bad.append((i, absolute, len(d), int(d[0]), int(d[-1]), float(np.abs(got - ref[absolute]).max()), got.ravel()[d[:4]].tolist(), ref[absolute].ravel()[d[:4]].tolist()))
                print('MISMATCH', bad[-1], flush=True)
        del junk
    for k in kids:
        k.kill()
    print(f'expand_stress: {a.iters} iterations, noise {a.noise}: {len(bad)} mismatching calls, {time.time() - t0:.1f} s')

if __name__ == '__main__':
    main()

"""Debug: where do the 50-step fp16 running sums of a full-size layer differ from the numpy oracle?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import heatmap_oracle as ho
from daam_amd.engine import HeatMapEngine

def to_bh(x, heads):
    b, s, c = x.shape
    d = c // heads
    return np.ascontiguousarray(x.reshape(b, s, heads, d).transpose(0, 2, 1, 3)).reshape(b * heads, s, d)

def ulp16(x):
    e = np.floor(np.log2(np.maximum(np.abs(x), 2.0 ** -14)))
    return 2.0 ** (e - 10)

heads, side, d, steps, n_q = 2, 64, 64, int(os.environ.get('STEPS', '50')), 5
hw = side * side
rng = np.random.default_rng(hw + d)
k = rng.standard_normal((2, 77, heads * d)).astype(np.float32); k[:, 0] *= 3.0; k = k.astype(np.float16)
qs = [rng.standard_normal((2, hw, heads * d)).astype(np.float16) for _ in range(n_q)]
k0 = k[1, 0].astype(np.float32).reshape(heads, d)
for q in qs:
    qq = q.reshape(2, hw, heads, d)
    qq[1] += (0.35 * k0 / np.sqrt((k0 ** 2).mean(-1, keepdims=True)))[None].astype(np.float16)
raw = ho.RawMaps(np.float16)
kb = to_bh(k, heads); qb = [to_bh(q, heads) for q in qs]
per_step = []
for s in range(steps):
    p = ho.tap(raw, 0, qb[s % n_q], kb, d ** -0.5, latent_hw=4096, pipe_dtype=np.float16)
    per_step.append(p)
want = np.stack([v for _, v in raw]).astype(np.float64)
for mode in ('fast', 'strict'):
    os.environ['DAAM_STRICT_EXP'] = '1' if mode == 'strict' else '0'
    for defer in (64, 0):
        eng = HeatMapEngine(1, tokens=77, out_side=64, accumulate='exact', defer_steps=defer)
        kd = torch.from_numpy(k).cuda(); qd = [torch.from_numpy(q).cuda() for q in qs]
        for s in range(steps):
            eng.tap_qk(0, qd[s % n_q], kd, heads, d ** -0.5, factor=1)
        got = torch.stack([v for _, v in eng.items()]).float().cpu().numpy().astype(np.float64)
        eng.close()
        diff = np.abs(got - want)
        ulps = diff / ulp16(np.maximum(np.abs(got), np.abs(want)))
        idx = np.unravel_index(np.argmax(ulps), ulps.shape)
        print(mode, 'defer', defer, 'max ulps', ulps.max(), 'at', idx, 'got', got[idx], 'want', want[idx], 'frac>0', (diff > 0).mean(),
              'frac>1ulp', (ulps > 1).mean(), 'frac>2', (ulps > 2).mean())
        h, t, y, x = idx
        # the per-step probabilities of that element in the oracle
        px = y * side + x
        pr = [float(per_step[s][heads + h, px, t]) for s in range(steps)]
        print('   oracle per-step probs:', ['%.3e' % v for v in pr[:10]])
        # one single step on the GPU for the same element
        eng = HeatMapEngine(1, tokens=77, out_side=64, accumulate='exact', defer_steps=0)
        eng.tap_qk(0, qd[0], kd, heads, d ** -0.5, factor=1)
        g1 = torch.stack([v for _, v in eng.items()]).float().cpu().numpy()
        eng.close()
        print('   gpu step-0 prob %.6e oracle %.6e' % (g1[idx], pr[0]))
        # histogram of ulps by magnitude of want
        for lo, hi in [(0, 2**-14), (2**-14, 2**-10), (2**-10, 2**-5), (2**-5, 1), (1, 100)]:
            m = (np.abs(want) >= lo) & (np.abs(want) < hi)
            if m.any():
                print('   |want| in [%g,%g): n=%d max ulps %.1f frac differing %.4f' % (lo, hi, m.sum(), ulps[m].max(), (diff[m] > 0).mean()))

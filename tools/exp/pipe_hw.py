"""Investigation: where do the slow waves of finalize_up32_pipe_kernel sit?  Per-wave loop durations next to HW_ID / XCC_ID.
    bash tools/exp/build_pipe_hw.sh && DAAM_HIP_LIB=tools/exp/libdaam_pipe_hw.so python tools/exp/pipe_hw.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from daam_amd.engine import HeatMapEngine
from daam_amd import _native as nat

dev = torch.device('cuda', 0)
layers = bench.topology('sdxl', 128)
sets = bench.make_inputs(layers, 2, dev, 1)
eng = HeatMapEngine(len(layers), defer_steps=4)
for t in range(4):
    for (layer, heads, side, d), (q, k) in zip(layers, sets[t % 2]):
        eng.tap_qk(layer, q, k, heads, d ** -0.5, factor=64 // side)
lib = nat.load()
hot_sets = bench.make_inputs(layers, 50, dev, 2)
hot_calls = bench.call_lists(layers, hot_sets, 64)
TAG = os.environ.get('PIPE_HW_TAG', 'base')
OUT = os.path.join(ROOT, 'gpurun_out')
os.makedirs(OUT, exist_ok=True)


def dump(tag, hot, **kw):
    if hot:
        for _ in range(12):
            bench.one_generation(eng, hot_calls, 50)
    for r in range(3):
        eng.global_heat_map(**kw)
    torch.cuda.synchronize()
    buf = np.zeros((4096, 12), dtype=np.uint64)
    lib.daam_debug_dump_pipe(buf.ctypes.data_as(ctypes.c_void_p))
    np.save(os.path.join(OUT, f'pipe_hw_{TAG}_{tag}.npy'), buf[:2002])
    b = buf[:2002]
    st = b[:, :6].astype(np.int64)
    st = (st - st[:, 0].min()) / 100.0                         # us
    loop = st[:, 4] - st[:, 3]
    hw, xcc = b[:, 8].astype(np.int64), b[:, 9].astype(np.int64) & 15
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    place = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    slot = place * 4 + simd
    wg = np.arange(2002) // 2
    print(f'== {tag}: span {st[:, 5].max():.1f} us; loop mean {loop.mean():.1f} min {loop.min():.1f} max {loop.max():.1f}; '
          f'percentiles 5/25/50/75/95: {np.round(np.percentile(loop, [5, 25, 50, 75, 95]), 1)}')
    print('  CUs used', len(np.unique(place)), ' SIMD slots used', len(np.unique(slot)), ' xcc ids', np.unique(xcc), ' se', np.unique(se), ' sh', np.unique(sh), ' cu', np.unique(cu))
    occ = np.bincount(slot, minlength=slot.max() + 1)[slot]    # waves that share this wave's SIMD over the kernel's life
    for o in np.unique(occ):
        m = occ == o
        print(f'  waves on a SIMD with {o} wave(s): {m.sum():5d}  loop mean {loop[m].mean():.1f} max {loop[m].max():.1f}  start mean {st[m, 0].mean():.2f}')
    same = simd[0::2] == simd[1::2]
    print(f'  workgroups whose two waves share a SIMD: {same.sum()} of 1001')
    per_cu = np.bincount(place, minlength=place.max() + 1)
    print('  waves per CU histogram', np.bincount(per_cu[per_cu > 0]))
    for x in np.unique(xcc):
        m = xcc == x
        print(f'  xcc {x}: waves {m.sum():4d} loop mean {loop[m].mean():.1f} max {loop[m].max():.1f} end max {st[m, 5].max():.1f} start mean {st[m, 0].mean():.2f}')
    chunk = wg // 77
    print('  by chunk (blockIdx.y): loop mean', np.round([loop[chunk == c].mean() for c in range(13)], 1))
    print('  by chunk: start mean', np.round([st[chunk == c, 0].mean() for c in range(13)], 2))
    slow = loop > np.percentile(loop, 95)
    print('  slowest 5%: xcc', np.bincount(xcc[slow], minlength=8), ' chunks', np.bincount(chunk[slow], minlength=13), ' occ', np.bincount(occ[slow]))
    # do the waves that share a SIMD overlap in time?  partner = the other wave in the same slot
    order = np.argsort(slot, kind='stable')
    pairs = [(order[i], order[i + 1]) for i in range(len(order) - 1) if slot[order[i]] == slot[order[i + 1]] and occ[order[i]] == 2]
    if pairs:
        a, c = np.array(pairs).T
        ov = np.minimum(st[a, 4], st[c, 4]) - np.maximum(st[a, 3], st[c, 3])
        print(f'  SIMD pairs: {len(pairs)}; loop overlap us mean {ov.mean():.1f} min {ov.min():.1f}; |start difference| mean {np.abs(st[a, 3] - st[c, 3]).mean():.1f}')
        tot = np.maximum(st[a, 4], st[c, 4]) - np.minimum(st[a, 3], st[c, 3])
        print(f'  pair busy time (first loop start -> last loop end) mean {tot.mean():.1f} max {tot.max():.1f}')


def timed(reps=30, **kw):
    nat.check(lib.daam_profile_enable(eng.ctx, 1))
    ts = []
    for r in range(reps + 3):
        eng.global_heat_map(**kw)
        ms = ctypes.c_float()
        nat.check(lib.daam_profile_last_ms(eng.ctx, 1, ctypes.byref(ms)))
        if r >= 3:
            ts.append(ms.value * 1e3)
    nat.check(lib.daam_profile_enable(eng.ctx, 0))
    return round(float(np.median(ts)), 1), round(float(np.min(ts)), 1)


print(TAG, 'finalize us (median, min): all keys', timed(), ' x2 keys only', timed(factors=[2]))
for _ in range(12):
    bench.one_generation(eng, hot_calls, 50)
print(TAG, 'HOT finalize us (median, min): all keys', timed(), ' x2 keys only', timed(factors=[2]))
np.save(os.path.join(OUT, f'fin_{TAG}_all.npy'), eng.global_heat_map().float().cpu().numpy())
np.save(os.path.join(OUT, f'fin_{TAG}_x2.npy'), eng.global_heat_map(factors=[2]).float().cpu().numpy())
for hot in (False, True):
    for name, kw in (('all', {}), ('x2', dict(factors=[2]))):
        dump(f'{name}_{"hot" if hot else "cold"}', hot, **kw)

"""Debug: per-wave phase timestamps of finalize_up32_pipe_kernel + HIP-event times of finalize for several key selections.
Needs a library built with -DDAAM_PIPE_TIMING (daam_amd.build.build_variant(out, ['-DDAAM_PIPE_TIMING']));
    DAAM_HIP_LIB=X.so python tools/exp/pipe_timing.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from daam_amd.engine import HeatMapEngine
from daam_amd import _native as nat
layers = bench.topology('sdxl', 128)
sets = bench.make_inputs(layers, 2, torch.device('cuda', 0), 1)
eng = HeatMapEngine(len(layers), defer_steps=4)
for t in range(4):
    for (layer, heads, side, d), (q, k) in zip(layers, sets[t % 2]):
        eng.tap_qk(layer, q, k, heads, d ** -0.5, factor=64 // side)
lib = nat.load()
HOT = bool(os.environ.get('PIPE_TIMING_HOT'))     # measure right behind back-to-back 50-step tap launches (the power-limited state of bench.py)
if HOT:
    hot_sets = bench.make_inputs(layers, 50, torch.device('cuda', 0), 2)
    hot_calls = bench.call_lists(layers, hot_sets, 64)


def heat():
    if HOT:
        for _ in range(12):
            bench.one_generation(eng, hot_calls, 50)



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


BRIEF = bool(os.environ.get('PIPE_TIMING_BRIEF'))
if not BRIEF:
    print('finalize us (median, min): all keys', timed(), ' x2 keys only', timed(factors=[2]), ' same-size only', timed(factors=[1]))
if hasattr(lib, 'daam_debug_dump_pipe'):
    for name, kw in (('all keys', {}), ('x2 only', dict(factors=[2]))):
        heat()
        for r in range(3):
            eng.global_heat_map(**kw)
        torch.cuda.synchronize()
        buf = np.zeros((4096, 8), dtype=np.uint64)
        rc = lib.daam_debug_dump_pipe(buf.ctypes.data_as(ctypes.c_void_p))
        cyc = (buf[:2002, 7].astype(np.int64) - buf[:2002, 6].astype(np.int64)).astype(np.float64)
        b = buf[:2002, :6].astype(np.int64)
        t0 = b[:, 0].min()
        b = (b - t0) * 10 / 1000.0   # us
        d = np.diff(b, axis=1)
        mhz = cyc / np.maximum(d[:, 3], 1e-3)                     # shader cycles per us of the loop = MHz while it ran
        print(f'  shader clock during the loop: median {np.median(mhz):.0f} MHz (min {mhz.min():.0f}, max {mhz.max():.0f}); '
              f'loop cycles per wave-iteration: median {np.median(cyc) / 78:.0f}')
        if BRIEF:
            print(f'{os.environ.get("DAAM_HIP_LIB", "default")[-12:]} {name}: span {b[:, 5].max():.1f} us; loop mean {d[:, 3].mean():.1f} min {d[:, 3].min():.1f} max {d[:, 3].max():.1f}; same-size phase mean {d[:, 1].mean():.1f}')
            continue
        print(f'--- {name}: kernel span us: {b[:, 5].max():.1f}')
        for i, nm in enumerate(['start', 'prefill issued', 'same-size done', 'ops loaded', 'loop done', 'end']):
            print(f'  {nm:16s} min {b[:, i].min():7.2f} mean {b[:, i].mean():7.2f} max {b[:, i].max():7.2f}')
        d = np.diff(b, axis=1)
        print('  phase durations us: mean', np.round(d.mean(0), 2), 'max', np.round(d.max(0), 2))

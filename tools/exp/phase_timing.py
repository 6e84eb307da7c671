"""Debug: per-phase wave latency of the tap kernel (needs tools/libdaam_ablate9.so, built with -DDAAM_ABLATE=9)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from daam_amd.engine import HeatMapEngine
from daam_amd import _native as nat
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
layers = bench.topology('sdxl', 128)
sets = bench.make_inputs(layers, 8, torch.device('cuda', 0), 1)
eng = HeatMapEngine(len(layers), defer_steps=steps)
for rep in range(2):
    for t in range(steps):
        for (layer, heads, side, d), (q, k) in zip(layers, sets[t % 8]):
            eng.tap_qk(layer, q, k, heads, d ** -0.5, factor=1)
    eng.flush()
torch.cuda.synchronize()
buf = np.zeros((1024, 8), dtype=np.uint64)
lib = nat.load()
print('rc', lib.daam_debug_dump(buf.ctypes.data_as(ctypes.c_void_p)))
tot = buf[:, :4].astype(np.float64) / (2 * steps)      # s_memtime ticks (100 MHz) per step
print('per-step wave latency in s_memtime ticks [commit+loop, barrier, lds+mfma, loads+softmax]:', tot.mean(0))
print('ns (10 ns per tick):', tot.mean(0) * 10, 'sum', tot.mean(0).sum() * 10)

"""Debug: per-workgroup phase timestamps of finalize_up32_mfma_kernel.  Needs a library built with
-DDAAM_FIN_TIMING:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DDAAM_FIN_TIMING daam_amd/csrc/*.hip -o X.so;
DAAM_HIP_LIB=X.so python tools/exp/fin_timing.py"""
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
        eng.tap_qk(layer, q, k, heads, d ** -0.5, factor=1)
for r in range(3):
    eng.global_heat_map()
torch.cuda.synchronize()
buf = np.zeros((1024, 4), dtype=np.uint64)
lib = nat.load()
print('rc', lib.daam_debug_dump_fin(buf.ctypes.data_as(ctypes.c_void_p)))
b = buf[:1001].astype(np.int64)
t0 = b[:, 0].min()
b = (b - t0) * 10 / 1000.0   # us
print('kernel span us:', b[:, 3].max())
for i, name in enumerate(['start', 'ops loaded', 'loop end', 'end']):
    print(f'{name:12s} min {b[:, i].min():7.2f} mean {b[:, i].mean():7.2f} max {b[:, i].max():7.2f}')
d = np.diff(b, axis=1)
print('phase durations us (prologue, loop, epilogue): mean', d.mean(0), 'max', d.max(0))
loop = d[:, 1]
wg = np.arange(1001)
tok, chunk = wg % 77, wg // 77
print('loop us by chunk:', [round(float(loop[chunk == c].mean()), 1) for c in range(13)])
print('loop us by wg % 8 (XCD):', [round(float(loop[wg % 8 == x].mean()), 1) for x in range(8)])
print('loop us by tok (first 16):', [round(float(loop[tok == t].mean()), 1) for t in range(16)])
order = np.argsort(loop)
print('slowest wgs:', [(int(w), round(float(loop[w]), 1)) for w in order[-8:]], 'fastest:', [(int(w), round(float(loop[w]), 1)) for w in order[:8]])

"""Debug: finalize time (HIP events around the finalize kernels, no clock monitor) for the bench workload."""
import ctypes, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from daam_amd.engine import HeatMapEngine
layers = bench.topology('sdxl', 128)
sets = bench.make_inputs(layers, 2, torch.device('cuda', 0), 1)
eng = HeatMapEngine(len(layers), defer_steps=4)
for t in range(4):
    for (layer, heads, side, d), (q, k) in zip(layers, sets[t % 2]):
        eng.tap_qk(layer, q, k, heads, d ** -0.5, factor=64 // side)
for _ in range(20):
    eng.global_heat_map()
ms = bench.measure_finalize(eng, reps=100)
print(json.dumps(dict(lib=os.environ.get('DAAM_HIP_LIB', 'default'), chunks=os.environ.get('DAAM_FIN_CHUNKS', 'default'),
                      paired=os.environ.get('DAAM_NO_PAIRED_FINALIZE', '0') != '1', finalize_us=round(ms * 1e3, 2))))

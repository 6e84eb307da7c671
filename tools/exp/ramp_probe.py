"""Debug: wall time of the first 40 generations of the bench workload, one sync each (context creation, clock ramp)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from daam_amd.engine import HeatMapEngine
dev = torch.device('cuda', 0)
layers = bench.topology('sdxl', 128)
sets = bench.make_inputs(layers, 50, dev, 1)
calls = bench.call_lists(layers, sets, 64)
eng = HeatMapEngine(len(layers), defer_steps=64)
torch.cuda.synchronize()
ts = []
for g in range(40):
    t0 = time.perf_counter()
    bench.one_generation(eng, calls, 50)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(' '.join(f'{t:.2f}' for t in ts))

"""Debug: what a fresh HeatMapEngine (a new trace) costs per generation compared with a reused one (MI355X)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from daam_amd.engine import HeatMapEngine
dev = torch.device('cuda', 0)
layers = bench.topology('sdxl', 128)
calls = bench.call_lists(layers, bench.make_inputs(layers, 2, dev, 1), 64)


def gen(eng, steps=2):
    eng.clear()
    for t in range(steps):
        for a in calls[t % 2]:
            eng.tap_qk(*a)
    return eng.global_heat_map()


eng = HeatMapEngine(len(layers), defer_steps=64)
gen(eng); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    gen(eng)
torch.cuda.synchronize()
reused = (time.perf_counter() - t0) / 10 * 1e3
eng.close()
t0 = time.perf_counter()
for _ in range(10):
    e = HeatMapEngine(len(layers), defer_steps=64)
    gen(e)
    torch.cuda.synchronize()
    e.close()
fresh = (time.perf_counter() - t0) / 10 * 1e3
print(f'2-step generation: reused engine {reused:.2f} ms, fresh engine each time {fresh:.2f} ms -> set-up + tear-down {fresh - reused:.2f} ms')

"""Debug: extraction time of a 10-step SDXL-1024 generation per pipeline dtype (fp16 = MFMA kernels, bf16 = any-shape kernels)."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from daam_amd.engine import HeatMapEngine
dev = torch.device('cuda', 0)
layers = bench.topology('sdxl', 128)
for dt in (torch.float16, torch.bfloat16):
    sets = [[(q.to(dt), k.to(dt)) for q, k in cur] for cur in bench.make_inputs(layers, 4, dev, 1)]
    calls = bench.call_lists(layers, sets, 64)
    eng = HeatMapEngine(len(layers), defer_steps=64)
    def gen(n=10):
        eng.clear()
        for t in range(n):
            for a in calls[t % 4]: eng.tap_qk(*a)
        eng.flush(); return eng.global_heat_map()
    gen(); torch.cuda.synchronize(); t0 = time.perf_counter(); gen(); gen(); torch.cuda.synchronize()
    print(dt, '%.2f ms per 10-step generation' % ((time.perf_counter() - t0) / 2 * 1e3))
    eng.close()

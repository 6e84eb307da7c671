"""Debug: where the host time of one bench generation goes (no device sync inside the loop)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from daam_amd.engine import HeatMapEngine
defer = int(sys.argv[1]) if len(sys.argv) > 1 else 16
layers = bench.topology('sdxl', 128)
sets = bench.make_inputs(layers, 8, torch.device('cuda', 0), 1)
calls = bench.call_lists(layers, sets, 64)
eng = HeatMapEngine(len(layers), defer_steps=defer)
for _ in range(3):
    bench.one_generation(eng, calls, 50)
torch.cuda.synchronize()
orig_flush = eng.flush
acc = dict(flush=0.0, clear=0.0, fin=0.0, n_flush=0)
def timed_flush():
    t = time.perf_counter(); orig_flush(); acc['flush'] += time.perf_counter() - t; acc['n_flush'] += 1
eng.flush = timed_flush
G = 10
t0 = time.perf_counter()
for g in range(G):
    t = time.perf_counter(); eng.clear(); acc['clear'] += time.perf_counter() - t
    tap = eng.tap_qk
    for s in range(50):
        for a in calls[s % len(calls)]:
            tap(*a)
    eng.flush()
    t = time.perf_counter(); eng.global_heat_map(); acc['fin'] += time.perf_counter() - t
host = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f'defer {defer}: host {host / G * 1e3:.3f} ms/gen (flush {acc["flush"] / G * 1e3:.3f} in {acc["n_flush"] / G:.0f} calls, '
      f'clear {acc["clear"] / G * 1e3:.3f}, global_heat_map {acc["fin"] / G * 1e3:.3f}); wall {wall / G * 1e3:.3f} ms/gen')

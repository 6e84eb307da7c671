"""What the SDXL tap launch owes to the FOOTPRINT of the recorded Q / K and what to their RE-USE (round 4).

profiles/r03_pool_sweep.json: the 50-step launch takes 1.65 ms while bench.py cycles <= 12 distinct step sets and 1.9-2.0 ms
from 25 on.  Two readings: (a) translation reach -- more distinct pages touched per launch; (b) re-use -- a pool of P sets
cycled over 50 steps means a workgroup re-reads its own Q tile / K rows every P steps, and with P <= 12 the bytes fetched in
between (P x 194 MB of conditional halves x the share of the chip's workgroups in flight) still sit in the 256 MB Infinity
Cache.  (b) is an artefact of cycling a pool; a real generation reads every tensor ONCE.  Cases that tell them apart, each one
HIP-event time of the tap launch (bench.measure_tap_kernel), per recorded step:

  pool 50 x 50 steps                 no re-use, 9.7 GB of conditional halves touched          (the bench default)
  pool 12 x 48 steps, interleaved    re-use distance 12 steps, 2.3 GB touched
  pool 12 x 48 steps, blocked        re-use distance 1 step (each set four times in a row), 2.3 GB touched
  pool 12 x 12 steps                 NO re-use inside a launch (the next launch re-reads it a whole launch later), 2.3 GB touched
  pool 50 x 12 steps                 no re-use, 12 of 50 sets per launch, all 50 over ~4 launches
  pool  3 x 12 steps                 re-use distance 3

(a) predicts pool12x12 < pool50x12 (per step); (b) predicts them equal and blocked <= interleaved < pool50.
``--arena``: every Q / K carved out of ONE allocation.   python tools/exp/footprint.py [--arena] -> gpurun_out/footprint.json"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from daam_amd.engine import HeatMapEngine  # noqa: E402


def main():
    arena = '--arena' in sys.argv[1:]
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    wl = bench.WORKLOADS['sdxl1024']
    layers = bench.topology(wl['kind'], wl['latent'])
    sets = bench.make_inputs(layers, 50, dev, seed=1234, arena=arena)
    calls = bench.call_lists(layers, sets, 64)
    eng = HeatMapEngine(len(layers), tokens=77, out_side=64, accumulate='exact', defer_steps=64,
                        defer_bytes=bench.default_defer_bytes(dev))
    for _ in range(12):                                       # clocks, code objects
        bench.one_generation(eng, calls, 50)
    cases = [('pool50 x 50', calls, 50), ('pool12 x 48 interleaved', calls[:12], 48),
             ('pool12 x 48 blocked', [calls[i // 4] for i in range(48)], 48), ('pool12 x 12', calls[:12], 12),
             ('pool50 x 12', calls, 12), ('pool3 x 12', calls[:3], 12), ('pool50 x 50 again', calls, 50)]
    _, qk, acc = bench.tap_bytes(layers, 1, 2, True)
    out = []
    for name, cl, steps in cases:
        mon = bench.ClockMonitor(eng, window_ms=40.0)
        ms = bench.measure_tap_kernel(eng, cl, steps, reps=12, fresh=True)
        clock = mon.read()
        rec = dict(case=name, arena=arena, steps=steps, tap_ms=round(ms, 4), us_per_step=round(ms * 1e3 / steps, 2),
                   us_per_step_less_sum_write=round((ms * 1e3 - acc / 5.0e6) / steps, 2),        # sums written at ~5 TB/s
                   gbs=round((steps * qk + acc) / ms / 1e6, 1), mhz=clock and clock['mhz_median_under_load'])
        out.append(rec)
        print(rec, file=sys.stderr, flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'footprint_arena.json' if arena else 'footprint.json'), 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

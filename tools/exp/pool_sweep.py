"""Tap launch time against the footprint of the recorded Q / K (bench.py --pool): the same kernel, the same bytes per launch,
more or fewer DISTINCT step sets resident in HBM.  ``python tools/exp/pool_sweep.py [pools...]`` -> gpurun_out/pool_sweep.json."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402


def main():
    arena = '--arena' in sys.argv[1:]                       # all step sets carved out of one allocation (bench.make_inputs)
    pools = [int(a) for a in sys.argv[1:] if a != '--arena'] or [3, 6, 12, 25, 50]
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    out = []
    for pool in pools:
        args = argparse.Namespace(pool=pool, defer_bytes=0, accumulate='exact', defer=64, arena=arena)
        r = bench.run_workload('sdxl1024', 50, 6, 2, dev, args, detail=False)
        out.append(dict(pool=pool, arena=arena, gb=round(pool * 0.388, 2), maps_per_s=round(r['value'], 1), tap_ms=round(r['roofline']['ms_per_launch'], 4),
                        hbm_frac=r['roofline']['frac'], clock=r['roofline_issue']['clock']['mhz_median_under_load'] if r['roofline_issue'] else None))
        print(out[-1], file=sys.stderr, flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'pool_sweep.json'), 'w'), indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

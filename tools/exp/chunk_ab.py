"""A / B of the chunked tap kernel (DAAM_TAP_CHUNKED=1, daam_amd/csrc/daam_tap_chunk.hip) against the default kernels, in ONE
process: bench.py's own workload leg (``run_workload``: heat maps / s, HIP-event time of the tap launch) with the switch off /
on / off / on.  ``python tools/exp/chunk_ab.py [sd15 sdxl1024 ...]`` -> one JSON line, also ``gpurun_out/chunk_ab.json``."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from daam_amd import engine as E  # noqa: E402


def main():
    names = sys.argv[1:] or ['sd15']
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    out = {}
    for name in names:
        wl = bench.WORKLOADS[name]
        denoise = wl.get('denoise_steps', 50)
        args = argparse.Namespace(pool=(6 if wl['kind'] == 'sdxl' else 0), defer_bytes=0, accumulate='exact', defer=64)
        gens = 40 if name == 'sd15' else 6
        runs = []
        for sw in ('0', '1', '0', '1'):
            os.environ['DAAM_TAP_CHUNKED'] = sw
            E.release_parked_contexts()
            r = bench.run_workload(name, denoise, gens, 2, dev, args, detail=False)
            runs.append(dict(chunked=int(sw), maps_per_s=round(r['value'], 1), tap_ms=round(r['roofline']['ms_per_launch'], 4),
                             hbm_frac=r['roofline']['frac'], kernels=r['roofline']['kernels_per_launch'],
                             side=r['roofline']['kernels_on_side_streams'], fin_ms=round(r['fin_ms'], 4)))
            print(name, runs[-1], file=sys.stderr, flush=True)
        out[name] = runs
    os.environ.pop('DAAM_TAP_CHUNKED', None)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'chunk_ab.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()

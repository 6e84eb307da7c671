#!/usr/bin/env python
"""List-scheduling model of the SD-v1.x deferred tap launch (tap_slab_kernel): which ORDER of the launch's workgroups empties
the chip earliest?  CPU only.

The launch is a few hundred to a thousand indivisible units (a workgroup = one (layer, 640-byte slab, pixel tile) walking all
recorded denoising steps: fp16 sums are order-dependent, a chain cannot be split in time) on 512 slots (256 CUs x 2 workgroups),
dispatched in workgroup-index order; every XCD takes an eighth of each segment of the order (daam_api.hip: daam_tap_flush).
Durations per unit kind are the measured ones (tools/exp/slab_timeline.py on a timing build: profiles/r0*_slab_timeline_sd15.json);
the model keeps them fixed, i.e. it ignores that a chain runs faster on a CU whose other slot is empty.

    python tools/slab_order_model.py [--half-us 92] [--profile profiles/r05_slab_timeline_sd15.json]

Prints the makespan of the shipped order, of every permutation of the segment order x tail share, and of the 'column' orders
(segments chosen so that every slot's units add up to the same length), and the perfect-packing bound (work / slots).
"""
import argparse
import heapq
import itertools
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# SD-v1.5 at 512 x 512: 15 hooked layers = 5 each of (head_dim 40, hw 4096), (80, 1024), (160, 256); 8 heads; a slab = 8 / 4 / 2 heads
LAYERS = {40: dict(n=5, hw=4096, slabs=1), 80: dict(n=5, hw=1024, slabs=2), 160: dict(n=5, hw=256, slabs=4)}
TILE = 32
SLOTS_PER_XCD, XCDS = 64, 8


def unit_counts(tail_pct):
    """units per kind: full 32-pixel tiles of each head_dim, and the 16-pixel half tiles of the head_dim-40 tail"""
    n = {}
    for d, l in LAYERS.items():
        tail_px = (l['hw'] * tail_pct // 100 // 32) * 32 if d == 40 and l['hw'] >= 64 else 0
        n[d] = l['n'] * l['slabs'] * ((l['hw'] - tail_px) // TILE)
        if d == 40:
            n['half'] = l['n'] * l['slabs'] * (tail_px // (TILE // 2))
    return n


def makespan(segments, dur):
    """segments: list of (kind, count); every XCD gets an eighth of each segment in order; in-order dispatch to the first free slot"""
    worst = 0.0
    for x in range(XCDS):
        queue = []
        for kind, count in segments:
            share = count // XCDS + (1 if x < count % XCDS else 0)
            queue += [dur[kind]] * share
        free = [0.0] * SLOTS_PER_XCD
        heapq.heapify(free)
        end = 0.0
        for t in queue:
            s = heapq.heappop(free)
            heapq.heappush(free, s + t)
            end = max(end, s + t)
        worst = max(worst, end)
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--profile', default=os.path.join(ROOT, 'profiles', 'r05_slab_timeline_sd15.json'))
    ap.add_argument('--half-us', type=float, default=0.0, help='duration of a 16-pixel head_dim-40 unit (0: 0.6 x the full unit -- same K slab, half the pixels)')
    ap.add_argument('--json', action='store_true')
    args = ap.parse_args()
    prof = json.load(open(args.profile))
    dur = {}
    for c in prof['classes']:
        loop = (c['loop_us_first_round'] or c['loop_us_later'])[1]
        dur[c['head_dim']] = loop + c['prologue_us'] + c['epilogue_us']
    dur['half'] = args.half_us or 0.6 * dur[40]
    out = dict(durations_us={str(k): round(v, 1) for k, v in dur.items()}, orders=[])

    def record(name, segments):
        work = sum(dur[k] * n for k, n in segments)
        out['orders'].append(dict(order=name, makespan_us=round(makespan(segments, dur), 1), perfect_us=round(work / (SLOTS_PER_XCD * XCDS), 1)))

    n25 = unit_counts(25)
    record('shipped: 160, 40, 80, half (tail 25 %)', [(160, n25[160]), (40, n25[40]), (80, n25[80]), ('half', n25['half'])])
    for tail in (0, 12, 25, 37, 50):
        n = unit_counts(tail)
        kinds = [160, 40, 80] + (['half'] if n['half'] else [])
        for perm in itertools.permutations(kinds):
            record(f'{", ".join(str(k) for k in perm)} (tail {tail} %)', [(k, n[k]) for k in perm])
    # column orders: the first 512 units are the first unit of every slot's column, the later segments follow in the order in which the
    # columns free up (short units first), the half tiles last: columns (40, 40), (80, 80, half), (160, 160, half)
    for tail in (12, 18, 25):
        n = unit_counts(tail)
        a = n[40] // 2
        record(f'columns: 40 x{a}, 80 x{n[80] // 2}, 160 x{n[160] // 2} | 160, 80, 40 | half (tail {tail} %)',
               [(40, a), (80, n[80] // 2), (160, n[160] // 2), (160, n[160] - n[160] // 2), (80, n[80] - n[80] // 2), (40, n[40] - a), ('half', n['half'])])
        record(f'columns, light first: 160 x{n[160] // 2}, 80 x{n[80] // 2}, 40 x{a} | 160, 80, 40 | half (tail {tail} %)',
               [(160, n[160] // 2), (80, n[80] // 2), (40, a), (160, n[160] - n[160] // 2), (80, n[80] - n[80] // 2), (40, n[40] - a), ('half', n['half'])])
    out['orders'].sort(key=lambda r: r['makespan_us'])
    if args.json:
        print(json.dumps(out))
        return
    print('durations (us):', out['durations_us'])
    shipped = next(r for r in out['orders'] if r['order'].startswith('shipped'))
    print(f'shipped order: {shipped["makespan_us"]} us (perfect packing {shipped["perfect_us"]})')
    for r in out['orders'][:12]:
        print(f'{r["makespan_us"]:7.1f} us  (perfect {r["perfect_us"]:6.1f})  {r["order"]}')


if __name__ == '__main__':
    main()

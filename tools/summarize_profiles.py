#!/usr/bin/env python
"""Condense rocprofv3 CSV output (gpurun_out/<dir>/...) into the small summaries committed under
profiles/:  kernel-trace stats of the libdaam_hip kernels, PMC means per kernel, and the HBM
traffic file bench.py reads (`profiles/hbm_traffic.json`).

    python tools/summarize_profiles.py --tag r01 --stats gpurun_out/prof_final \
        --pmc gpurun_out/pmc_sq gpurun_out/pmc_fetch gpurun_out/pmc_write --key sdxl1024:defer8:exact

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md (section HBM): FETCH_SIZE / WRITE_SIZE are in KiB;
on gfx950 FETCH_SIZE reports half of a wide coalesced read stream, so reads = 2 * FETCH_SIZE * 1024.
(Check: finalize_same / finalize_up read exactly their 63 MB / 158 MB of planes by this rule.)"""
import argparse
import collections
import csv
import glob
import json
import os
import statistics


def find(d, suffix):
    fs = glob.glob(os.path.join(d, '**', f'*{suffix}'), recursive=True)
    return fs[0] if fs else None


def short(name):
    if 'daam' not in name:
        return None
    for k in ('tap_d64_kernel', 'tap_slab_kernel', 'tap_chunk_kernel', 'tap_wide_kernel', 'tap_mfma_kernel', 'tap_generic_kernel', 'tap_probs_kernel', 'attend_kernel', 'finalize_up32_pipe_kernel',
              'finalize_up32_same_kernel', 'finalize_up32_mfma_kernel', 'finalize_down2_kernel',
              'finalize_up_kernel', 'finalize_same_kernel', 'finalize_kernel', 'normalize_kernel', 'word_'):
        if k in name:
            return k
    return name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', required=True)
    ap.add_argument('--stats')
    ap.add_argument('--pmc', nargs='*', default=[])
    ap.add_argument('--key', default='sdxl1024:defer8:exact')
    ap.add_argument('--out', default='profiles')
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    if a.stats:
        f = find(a.stats, 'kernel_stats.csv')
        rows = list(csv.DictReader(open(f)))
        keep = [r for r in rows if short(r['Name'])] + [r for r in rows if not short(r['Name'])][:4]
        with open(os.path.join(a.out, f'{a.tag}_kernel_stats.csv'), 'w', newline='') as o:
            w = csv.DictWriter(o, fieldnames=list(rows[0].keys()))
            w.writeheader()
            for r in keep:
                r = dict(r)
                r['Name'] = r['Name'][:120]
                w.writerow(r)
        print('wrote', os.path.join(a.out, f'{a.tag}_kernel_stats.csv'))
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in a.pmc:
        f = find(d, 'counter_collection.csv')
        if not f:
            continue
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k:
                pmc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    if pmc:
        summary = {k: {c: dict(n=len(v), mean=statistics.fmean(v), median=statistics.median(v), max=max(v))
                       for c, v in cs.items()} for k, cs in pmc.items()}
        json.dump(summary, open(os.path.join(a.out, f'{a.tag}_pmc_summary.json'), 'w'), indent=1, sort_keys=True)
        print('wrote', os.path.join(a.out, f'{a.tag}_pmc_summary.json'))
        tpath = os.path.join(a.out, 'hbm_traffic.json')
        traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
        rec = {}
        for kern, field in (('tap_d64_kernel', 'tap'), ('tap_slab_kernel', 'tap'), ('tap_chunk_kernel', 'tap'), ('tap_mfma_kernel', 'tap'), ('finalize_up32_pipe_kernel', 'finalize_pipe'),
                            ('finalize_up32_same_kernel', 'finalize_pair'), ('finalize_up32_mfma_kernel', 'finalize_up'),
                            ('finalize_same_kernel', 'finalize_same')):
            cs = pmc.get(kern, {})
            if f'{field}_bytes_per_launch' in rec:
                continue
            if 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs:
                # the steady-state launch = the most common large one: take the upper median
                fe = statistics.median(sorted(cs['FETCH_SIZE'])[len(cs['FETCH_SIZE']) // 2:])
                wr = statistics.median(sorted(cs['WRITE_SIZE'])[len(cs['WRITE_SIZE']) // 2:])
                rec[f'{field}_bytes_per_launch'] = int((2 * fe + wr) * 1024)
                rec[f'{field}_read_bytes'] = int(2 * fe * 1024)
                rec[f'{field}_write_bytes'] = int(wr * 1024)
        if rec:
            rec['method'] = 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes); bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024'
            traffic[a.key] = rec
            json.dump(traffic, open(tpath, 'w'), indent=1, sort_keys=True)
            print('wrote', tpath, rec)
        # what bench.py reads for its rooflines (profiles/<tag>_counters.json), tied to the kernel sources by their fingerprint
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from daam_amd.build import csrc_sha, kernel_shas
        cpath = os.path.join(a.out, a.tag.split('_')[0] + '_counters.json')      # one file per round, one entry per workload
        counters = json.load(open(cpath)) if os.path.exists(cpath) else {}
        if counters.get('csrc_sha') != csrc_sha():
            counters = dict(csrc_sha=csrc_sha(), workloads={}, kernel_shas=dict(sorted(kernel_shas().items())),
                            kernel_shas_note='machine-code fingerprints of every kernel of the measured build (daam_amd.build.kernel_shas): bench.py '
                                             'accepts this file for a later build whose sources differ while all of these are byte-identical in it',
                            method='rocprofv3 --pmc passes of `python bench.py --workload W` (tools/profile_round.sh); per launch: '
                                   'upper-median over the launches of a kernel; *_per_simd = counter / 1024 SIMDs; '
                                   'VALU busy cycles = SQ_ACTIVE_INST_VALU (quad-cycles) x 4; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024')
        w = dict(rec) if rec else {}
        w.pop('method', None)

        def upper_median(v):
            v = sorted(v)
            return statistics.median(v[len(v) // 2:])
        tap_names = [k for k in ('tap_d64_kernel', 'tap_slab_kernel', 'tap_chunk_kernel', 'tap_wide_kernel', 'tap_mfma_kernel') if k in pmc and 'SQ_ACTIVE_INST_VALU' in pmc[k]]
        if tap_names:
            w['tap_kernels_per_launch'] = len(tap_names)          # bench.py drops the tap counters when the launch structure differs
            # a flush may run several tap kernels side by side (SD-v1.5): their work adds up on the same SIMDs
            w['tap_valu_busy_cycles_per_simd'] = round(sum(upper_median(pmc[k]['SQ_ACTIVE_INST_VALU']) for k in tap_names) * 4 / 1024, 1)
            w['tap_valu_insts_per_simd'] = round(sum(upper_median(pmc[k]['SQ_INSTS_VALU']) for k in tap_names) / 1024, 1)
            if all('SQ_INSTS_MFMA' in pmc[k] for k in tap_names):
                w['tap_mfma_per_simd'] = round(sum(upper_median(pmc[k]['SQ_INSTS_MFMA']) for k in tap_names) / 1024, 1)
            if all('SQ_VALU_MFMA_BUSY_CYCLES' in pmc[k] for k in tap_names):
                w['tap_mfma_busy_cycles_per_simd'] = round(sum(upper_median(pmc[k]['SQ_VALU_MFMA_BUSY_CYCLES']) for k in tap_names) / 1024, 1)
        fin_names = [k for k in ('finalize_up32_pipe_kernel', 'finalize_up32_same_kernel', 'finalize_up32_mfma_kernel', 'finalize_same_kernel', 'finalize_down2_kernel', 'finalize_up_kernel',
                                 'finalize_kernel') if k in pmc and 'SQ_ACTIVE_INST_VALU' in pmc[k]]
        if fin_names:
            w['finalize_valu_busy_cycles_per_simd'] = round(sum(upper_median(pmc[k]['SQ_ACTIVE_INST_VALU']) for k in fin_names) * 4 / 1024, 1)
            w['finalize_mfma_per_simd'] = round(sum(upper_median(pmc[k].get('SQ_INSTS_MFMA', [0])) for k in fin_names) / 1024, 1)
            w['finalize_kernels'] = fin_names
            fe = sum(upper_median(pmc[k]['FETCH_SIZE']) for k in fin_names if 'FETCH_SIZE' in pmc[k])
            wr = sum(upper_median(pmc[k]['WRITE_SIZE']) for k in fin_names if 'WRITE_SIZE' in pmc[k])
            if fe:
                w['finalize_bytes_per_launch'] = int((2 * fe + wr) * 1024)
        counters['workloads'][a.key] = w
        json.dump(counters, open(cpath, 'w'), indent=1, sort_keys=True)
        print('wrote', cpath, w)


if __name__ == '__main__':
    main()

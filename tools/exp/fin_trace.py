#!/usr/bin/env python
"""Timeline of the finalize launches in a rocprofv3 --kernel-trace CSV: start / end of the upload, class kernels of the last
few compute_global_heat_map calls relative to the upload kernel (which kernel waits for which, how long the gaps are).
    python tools/exp/fin_trace.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    names = [r['Kernel_Name'] for r in rows]
    idx = [i for i, n in enumerate(names) if 'finalize_up32_pipe' in n or 'finalize_up32_same' in n]
    for i in idx[-6:]:
        lo = i
        while lo > 0 and 'upload_kernel' not in names[lo]:
            lo -= 1
        t0 = int(rows[lo]['Start_Timestamp'])
        hi = min(len(rows), i + 3)
        print('--- finalize call')
        for r in rows[lo:hi]:
            n = r['Kernel_Name']
            if 'daam' not in n:
                continue
            short = n.split('(')[0].split('::')[-1][:34]
            print(f'  {short:36s} start {(int(r["Start_Timestamp"]) - t0) / 1e3:8.1f} us  end {(int(r["End_Timestamp"]) - t0) / 1e3:8.1f} us  '
                  f'dur {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:7.1f}  grid {r.get("Grid_Size", "?")} wg {r.get("Workgroup_Size", "?")} q {r.get("Queue_Id", "?")}')


if __name__ == '__main__':
    main()

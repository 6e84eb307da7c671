#!/usr/bin/env python
"""Which host do the committed golden fixtures belong to, and what changes on another one?   (build container only: needs /root/reference)

    python tools/golden_host_check.py            -> tests/golden/PROVENANCE.json

Regenerates every ``tests/golden/<case>.npz`` with ``oracle/make_golden.py`` (the UNMODIFIED reference) on THIS host into a scratch
directory and compares array by array with the committed fixture: bit-identical, or -- the literal fp16 / bf16 cases on a CPU other than
the one the fixture came from -- how many elements differ and by how many ulps of the array's dtype.  The committed fixtures of rounds
1-4 did not record their host; this file records, per case, the host of the check and the outcome, so that a reader can tell "same bits
as on host X" from "within the flip class of host X"."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def grid_ulps(a, b, dtype):
    """largest difference in units of the last place of the PIPELINE dtype (the arrays are stored widened to f32)"""
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    diff = np.abs(a64 - b64)
    if not diff.any():
        return 0.0
    mag = np.maximum(np.maximum(np.abs(a64), np.abs(b64)), 2.0 ** -24)
    mant = {'float16': 10, 'bfloat16': 7, 'float32': 23}[dtype]
    ulp = 2.0 ** (np.floor(np.log2(mag)) - mant)
    if dtype == 'float16':
        ulp = np.maximum(ulp, 2.0 ** -24)                       # subnormals
    return float((diff / ulp).max())


def main():
    from oracle import make_golden as mg
    cases = list(mg.CASES)
    if len(sys.argv) > 2 and sys.argv[1] == '--compare-only':
        scratch = sys.argv[2]
    else:
        scratch = tempfile.mkdtemp(prefix='golden_check_')
        env = dict(os.environ, DAAM_GOLDEN_OUT=scratch)
        subprocess.run([sys.executable, '-m', 'oracle.make_golden', *cases], cwd=ROOT, env=env, check=True)
    import torch  # noqa: F401  (host_info)
    report = dict(checked_on=mg.host_info(),
                  note='every case regenerated with oracle/make_golden.py (the unmodified reference) on the host above and compared with the committed '
                       'fixture.  Fixtures of rounds 1-4 carry no host record of their own (meta.host exists from round 5 on): this file is what ties '
                       'them to a host.  Classes: raw = running sums (raw_*), out = attention outputs of the processor (out_rows_*), maps = global / word '
                       'maps (f32); ulps are in units of the last place of the PIPELINE dtype', cases={})
    for name in cases:
        old = np.load(os.path.join(ROOT, 'tests', 'golden', f'{name}.npz'), allow_pickle=False)
        new = np.load(os.path.join(scratch, f'{name}.npz'), allow_pickle=False)
        dtype = json.loads(str(old['meta']))['dtype']
        rec = dict(dtype=dtype, arrays=len(old.files) - 1, identical=0)
        cls = {}
        for k in old.files:
            if k == 'meta':
                continue
            a, b = old[k], new[k]
            same = a.shape == b.shape and a.dtype == b.dtype and (np.array_equal(a, b, equal_nan=True) if a.dtype.kind in 'fc' else np.array_equal(a, b))
            if same:
                rec['identical'] += 1
                continue
            kind = 'raw' if k.startswith('raw_') else 'out' if k.startswith('out_rows') else 'maps' if k.startswith(('global', 'word')) else 'sums'
            c = cls.setdefault(kind, dict(arrays=0, elements_differing=0, elements=0, max_abs=0.0, max_ulps=0.0))
            c['arrays'] += 1
            if a.shape != b.shape or a.dtype.kind != 'f':
                c['shape_or_type_changed'] = True
                continue
            c['elements_differing'] += int((a != b).sum())
            c['elements'] += int(a.size)
            c['max_abs'] = max(c['max_abs'], float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()))
            if kind in ('raw', 'out'):
                c['max_ulps'] = max(c['max_ulps'], round(grid_ulps(a, b, dtype), 2))
        for c in cls.values():
            c['share'] = round(c['elements_differing'] / max(c['elements'], 1), 6)
        rec['differing'] = cls
        rec['verdict'] = 'bit-identical on this host' if not cls else 'differs in the last bits (another host\'s reduced-precision GEMM order): see the classes'
        report['cases'][name] = rec
        print(name, rec['verdict'], {k: (v['share'], v['max_ulps'], v['max_abs']) for k, v in cls.items()}, flush=True)
    json.dump(report, open(os.path.join(ROOT, 'tests', 'golden', 'PROVENANCE.json'), 'w'), indent=1)
    print('wrote tests/golden/PROVENANCE.json')


if __name__ == '__main__':
    main()

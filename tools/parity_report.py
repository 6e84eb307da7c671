"""Observed deviation of the HIP path from the golden vectors of the unmodified reference
(tests/golden/*.npz), per case and tap mode.  Run on an MI355X:  python tools/parity_report.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
from conftest import GOLDEN_CASES, golden_pipe, load_golden
import daam_amd

rows = []
for name in GOLDEN_CASES:
    z, meta = load_golden(name)
    for tap, defer in (('qk', 0), ('qk', 8), ('probs', 0)):
        pipe = golden_pipe(meta, device='cuda:0')
        with daam_amd.trace(pipe, tap=tap, defer_steps=defer) as tc:
            pipe(meta['prompt'], num_inference_steps=meta['steps'])
            items = list(tc.all_heat_maps)
            raw_err = 0.0
            for sid in z['raw_sample_ids']:
                got = items[int(sid)][1][[0, 1, 2, 76]].float().cpu().numpy()
                raw_err = max(raw_err, float(np.abs(got - z[f'raw_{int(sid)}']).max()))
            # plain maps: absolute error; normalize=True variants divide by a sum that can be tiny (values up to
            # 1e5 in the no-CFG / batch-4 cases), so they are reported relative to the largest reference value
            g_err, g_rel = 0.0, 0.0
            for vn, kw in json.loads(str(z['variants'])).items():
                got = tc.compute_global_heat_map(**kw).heat_maps.cpu().numpy()
                want = z[f'global_{vn}']
                err = float(np.abs(got - want).max())
                if not kw.get('normalize'):
                    g_err = max(g_err, err)
                g_rel = max(g_rel, err / max(1.0, float(np.abs(want).max())))
        rows.append(dict(case=name, dtype=meta['dtype'], tap=tap, defer=defer, raw_sum_max_abs=raw_err,
                         global_map_max_abs=g_err, global_map_max_rel_all_variants=g_rel))
        print(rows[-1], flush=True)
out = dict(softmax='compensated (DAAM_STRICT_EXP=1)' if os.environ.get('DAAM_STRICT_EXP') == '1' else 'fast (default)', rows=rows)
print(json.dumps(out))

"""The numpy oracle (oracle/heatmap_oracle.py) against the golden vectors produced by
executing the unmodified reference (oracle/make_golden.py).  CPU only."""
import json

import numpy as np
import pytest

from conftest import golden_pipe
from oracle import heatmap_oracle as ho
from oracle.make_golden import SAMPLE_TOKENS, input_checksums


def _tol(meta):
    # fp32: differences are summation-order only.  fp16: identical rounding points; a
    # straddled fp16 rounding in a logit moves one probability by 1 ulp (<= 4.9e-4) once.
    # bf16: same, with 8-bit significands (1 ulp <= 3.9e-3).
    return {'float32': 2e-6, 'float16': 2e-4, 'bfloat16': 3e-3}[meta['dtype']]


def test_oracle_matches_reference(golden_case):
    name, z, meta = golden_case
    pipe = golden_pipe(meta)
    np.testing.assert_allclose(input_checksums(pipe, meta['steps']), z['input_checksums'], rtol=0, atol=0)
    import torch
    dt = getattr(torch, meta['dtype'])
    raw = ho.replay_generation(pipe, meta['steps'], dt, locate_middle_block=bool(meta.get('heads')))
    keys = np.asarray([k for k, _ in raw], dtype=np.int32)
    np.testing.assert_array_equal(keys, z['keys'])          # same keys, same insertion order
    sums = np.asarray([float(v.astype(np.float64).sum()) for _, v in raw])
    np.testing.assert_allclose(sums, z['key_sum'], rtol={'float32': 1e-5, 'float16': 1e-4, 'bfloat16': 2e-3}[meta['dtype']])
    items = list(raw)
    # EVERY (key, token) plane, by two checksums (plain and position-weighted: a transposed / shifted plane changes the second)
    rt = {'float32': 1e-5, 'float16': 1e-3, 'bfloat16': 8e-3}[meta['dtype']]
    ps = np.stack([v.astype(np.float64).sum((1, 2)) for _, v in items])
    pw = np.stack([(v.astype(np.float64) * np.arange(1, v.shape[1] * v.shape[2] + 1, dtype=np.float64).reshape(1, v.shape[1], v.shape[2])).sum((1, 2))
                   for _, v in items])
    np.testing.assert_allclose(ps, z['plane_sum'], rtol=rt, atol=rt)
    np.testing.assert_allclose(pw, z['plane_wsum'], rtol=rt, atol=rt * 1e3)
    for sid in z['raw_sample_ids']:
        got = items[int(sid)][1][SAMPLE_TOKENS].astype(np.float32)
        want = z[f'raw_{int(sid)}']
        # fp16 running sums: one flipped rounding = 1 ulp of the sum (SOS sums reach ~steps)
        ulp = {'float16': 2.0 ** -10, 'bfloat16': 2.0 ** -7}.get(meta['dtype'])
        tol = _tol(meta) if ulp is None else ulp * max(1.0, float(want.max()))
        np.testing.assert_allclose(got, want, rtol=0, atol=tol)
    lat = ho.latent_hw_for(pipe.unet.config.sample_size, pipe.vae_scale_factor)
    n_rows = len(pipe.tokenizer.tokenize(meta['prompt'])) + 2
    variants = json.loads(str(z['variants']))
    for vn, kw in variants.items():
        got = ho.global_heat_map(raw, lat, n_rows=n_rows, **kw)
        want = z[f'global_{vn}']
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=_tol(meta) * max(1.0, float(np.abs(want).max())),
                                   err_msg=f'{name}:{vn}')


def test_oracle_processor_outputs_match_reference(golden_case):
    """``ho.attention_output`` (+ ``batch_to_head_dim`` + the output projection): what the reference's processor RETURNED
    for every cross-attention call of the last step (trace.py:296-304), against the rows and sums recorded when the
    unmodified reference ran."""
    name, z, meta = golden_case
    if 'out_sums' not in z or meta['dtype'] == 'bfloat16':
        pytest.skip('case carries no processor outputs')
    import torch
    from oracle.make_golden import OUT_SAMPLE_ROWS
    pipe = golden_pipe(meta)
    outs = []
    ho.replay_generation(pipe, meta['steps'], getattr(torch, meta['dtype']), locate_middle_block=bool(meta.get('heads')),
                         outputs=outs)
    assert len(outs) == len(z['out_sums'])
    rel = 1e-5 if meta['dtype'] == 'float32' else 2e-3
    for i, o in enumerate(outs):
        want = z[f'out_rows_{i}']
        got = o[-1, :OUT_SAMPLE_ROWS].astype(np.float32)
        assert np.abs(got - want).max() <= rel * max(float(np.abs(want).max()), 1e-6), f'{name}: call {i}'
        sq = float((o.astype(np.float64) ** 2).sum())
        assert abs(sq - z['out_sums'][i, 1]) <= 4 * rel * z['out_sums'][i, 1], f'{name}: call {i}'


def test_oracle_word_heat_map(golden_case):
    name, z, meta = golden_case
    pipe = golden_pipe(meta)
    gm = z['global_default']
    word = str(z['word'])
    toks = pipe.tokenizer.tokenize(meta['prompt'].lower())
    idxs, _ = ho.token_merge_indices(toks, pipe.tokenizer.tokenize(word.lower()), word)
    wm = ho.word_heat_map(gm, idxs)
    np.testing.assert_allclose(wm, z['word_map'], rtol=0, atol=1e-6)
    np.testing.assert_allclose(ho.expand_as(wm, 128), z['word_expand_128'], rtol=0, atol=2e-5)
    np.testing.assert_allclose(ho.expand_as(wm, 128, absolute=True), z['word_expand_128_abs'], rtol=0, atol=2e-5)


@pytest.mark.parametrize('i,o', [(32, 64), (16, 64), (64, 64), (128, 64), (64, 128), (24, 96), (64, 512)])
def test_bicubic_matches_torch(i, o):
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(i * 1000 + o)
    x = rng.standard_normal((3, i, i)).astype(np.float32)
    want = F.interpolate(torch.from_numpy(x)[:, None], size=(o, o), mode='bicubic')[:, 0].numpy()
    got = ho.bicubic_resize(x, o)
    np.testing.assert_allclose(got, want, rtol=0, atol=3e-6)


def test_no_maps_errors():
    raw = ho.RawMaps()
    with pytest.raises(RuntimeError, match='Did you forget'):
        ho.global_heat_map(raw, 4096)
    with pytest.raises(RuntimeError, match='given parameters'):
        ho.global_heat_map(raw, 4096, head_idx=3)


def test_oracle_iou_ioa_match_reference():
    """oracle compute_iou / compute_ioa (evaluate.py:14-35 restated) against the reference's own results
    (tests/golden/evaluate.npz, made by oracle/make_golden.py from the unmodified daam/evaluate.py)."""
    import os
    from conftest import GOLDEN_DIR
    z = np.load(os.path.join(GOLDEN_DIR, 'evaluate.npz'))
    for name in json.loads(str(z['names'])):
        a, b = z[f'{name}_a'], z[f'{name}_b']
        iou = np.asarray([ho.compute_iou(a[i], b[i]) for i in range(len(a))])
        ioa = np.asarray([ho.compute_ioa(a[i], b[i]) for i in range(len(a))])
        tol = 0.0 if 'binary' in name else 2e-7          # binary masks: integer sums, exact in any order
        np.testing.assert_allclose(iou, z[f'{name}_iou'], rtol=0, atol=tol, err_msg=name)
        np.testing.assert_allclose(ioa, z[f'{name}_ioa'], rtol=0, atol=tol, err_msg=name)


def test_golden_provenance_is_recorded():
    """tests/golden/PROVENANCE.json (tools/golden_host_check.py): every case regenerated on a named host and compared with the
    committed fixture.  The fp32 cases must be bit-identical on any host; the literal fp16 / bf16 cases may differ from another
    host's in the last bit of a small share of the running sums (another CPU's reduced-precision GEMM order) -- the class of
    deviation the GPU tolerances are built around."""
    import json
    import os
    from conftest import GOLDEN_CASES, GOLDEN_DIR
    rec = json.load(open(os.path.join(GOLDEN_DIR, 'PROVENANCE.json')))
    assert rec['checked_on']['cpu'] and rec['checked_on']['torch']
    assert set(rec['cases']) == set(GOLDEN_CASES)
    for name, c in rec['cases'].items():
        if c['dtype'] == 'float32':
            assert c['verdict'].startswith('bit-identical'), (name, c['verdict'])
        else:
            raw = c['differing'].get('raw')
            if raw:                                               # running sums: at most a couple of ulps, in well under 1 % of the elements
                assert raw['share'] < 0.01 and raw['max_ulps'] <= 4, (name, raw)
            maps = c['differing'].get('maps')
            if maps:
                assert maps['max_abs'] < 1e-3, (name, maps)

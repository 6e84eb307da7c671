"""``compute_iou`` / ``compute_ioa`` (reference daam/evaluate.py:14-35) and
``WordHeatMap.compute_ioa`` (daam/heatmap.py:95-96) on the HIP kernel ``daam_mask_overlap``, against the golden results of
the unmodified reference (tests/golden/evaluate.npz) and the numpy oracle.  Run with ``-m gpu`` on an MI355X."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import heatmap_oracle as ho

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _golden():
    z = np.load(os.path.join(GOLDEN_DIR, 'evaluate.npz'))
    return z, json.loads(str(z['names']))


def test_iou_ioa_match_reference_golden():
    import daam_amd
    from daam_amd.evaluate import compute_ioa_batch, compute_iou_batch, mask_overlap
    z, names = _golden()
    for name in names:
        a, b = torch.from_numpy(z[f'{name}_a']).to(DEV), torch.from_numpy(z[f'{name}_b']).to(DEV)
        # power-of-two upscales of binary masks and same-size binary masks: every sum is an exact integer -> the fp32 ratio
        # is the reference's to the last bit; soft predictions: the bicubic value next to the threshold / the summation order
        # of non-integer sums may differ in the last place
        tol = 0.0 if 'binary' in name else 1e-6
        iou, ioa = compute_iou_batch(a, b), compute_ioa_batch(a, b)
        np.testing.assert_allclose(iou, z[f'{name}_iou'].astype(np.float32), rtol=0, atol=tol, err_msg=name)
        np.testing.assert_allclose(ioa, z[f'{name}_ioa'].astype(np.float32), rtol=0, atol=tol, err_msg=name)
        # one pair at a time = the reference's call signature
        for i in range(a.shape[0]):
            assert abs(daam_amd.compute_iou(a[i], b[i]) - float(z[f'{name}_iou'][i])) <= max(tol, 1e-7)
            assert abs(daam_amd.compute_ioa(a[i], b[i]) - float(z[f'{name}_ioa'][i])) <= max(tol, 1e-7)
        # the three sums against the oracle
        sums = mask_overlap(a, b).cpu().numpy()
        for i in range(a.shape[0]):
            want = ho.mask_overlap(z[f'{name}_a'][i], z[f'{name}_b'][i])
            np.testing.assert_allclose(sums[i], np.asarray(want), rtol=2e-6 if 'soft' in name else 0, atol=0)


@pytest.mark.parametrize('shape', [((64, 64), (512, 512)), ((64, 64), (1024, 1024)), ((17, 23), (40, 31)), ((96, 96), (64, 64))])
def test_mask_overlap_vs_oracle_random_shapes(shape):
    """Odd sizes, non-square masks, down-scaling: the sums against the oracle; pixels whose bicubic value lies within 1e-5 of the
    threshold may fall on either side (the reference itself is platform-dependent there)."""
    from daam_amd.evaluate import mask_overlap
    (ah, aw), (bh, bw) = shape
    rng = np.random.default_rng(ah * 1000 + bw)
    a = (rng.random((2, ah, aw)) * 2.0).astype(np.float32)
    b = (rng.random((2, bh, bw)) > 0.5).astype(np.float32)
    got = mask_overlap(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)).cpu().numpy()
    for i in range(2):
        want = np.asarray(ho.mask_overlap(a[i], b[i]))
        assert np.abs(got[i] - want).max() <= 1e-4 * bh * bw + 1e-3, (shape, got[i], want)


def test_word_heat_map_compute_ioa_and_errors():
    import daam_amd
    from daam_amd.heatmap import WordHeatMap
    rng = np.random.default_rng(3)
    x, y = rng.random((64, 64)).astype(np.float32), rng.random((64, 64)).astype(np.float32)
    got = WordHeatMap(torch.from_numpy(x).to(DEV), 'a').compute_ioa(WordHeatMap(torch.from_numpy(y).to(DEV), 'b'))
    assert abs(got - ho.compute_ioa(x, y)) <= 1e-6                      # same shapes: no resize, no binarisation (evaluate.py:27)
    # CPU masks (what the reference's evaluation flow passes: load_mask / expand_as(...).cpu()) are moved to the HIP device
    a_cpu, b_cpu = torch.from_numpy(x), (torch.from_numpy(y) > 0.5).float()
    # (soft prediction: the f32 atomics of the three sums add in another order from launch to launch -> 1e-6)
    assert abs(daam_amd.compute_iou(a_cpu, b_cpu) - daam_amd.compute_iou(a_cpu.to(DEV), b_cpu.to(DEV))) <= 1e-6
    assert abs(daam_amd.compute_ioa(a_cpu, b_cpu.to(DEV)) - daam_amd.compute_ioa(a_cpu.to(DEV), b_cpu.to(DEV))) <= 1e-6
    assert abs(daam_amd.compute_iou(a_cpu, b_cpu) - ho.compute_iou(x, (y > 0.5).astype(np.float32))) <= 1e-6
    with pytest.raises(daam_amd._native.DaamError):
        daam_amd.compute_iou(torch.zeros(8, 4, device=DEV), torch.zeros(8, 6, device=DEV))   # same height, other width


def test_nan_prediction_gives_nan_like_the_reference():
    """evaluate.py:17-18 are two masked assignments: a NaN in the resized prediction satisfies neither, stays a NaN and makes
    the ratio NaN (round-2 advisor: the kernel used to turn it into 1)."""
    import daam_amd
    a = torch.ones(8, 8)
    a[3, 4] = float('nan')
    b = (torch.rand(32, 32, generator=torch.Generator().manual_seed(1)) > 0.5).float()
    assert np.isnan(daam_amd.compute_iou(a.to(DEV), b.to(DEV))) and np.isnan(daam_amd.compute_ioa(a.to(DEV), b.to(DEV)))
    assert np.isnan(ho.compute_iou(a.numpy(), b.numpy()))
    a[3, 4] = 1.0
    assert abs(daam_amd.compute_iou(a.to(DEV), b.to(DEV)) - ho.compute_iou(a.numpy(), b.numpy())) <= 1e-6


def test_evaluators_on_the_device():
    """MeanEvaluator / UnsupervisedEvaluator (evaluate.py:46-117) over the kernel: the best candidate of a list scored in one
    launch is the reference's max over ``compute_iou`` calls (golden IoUs), CPU masks and ragged candidate lists included."""
    from daam_amd.evaluate import MeanEvaluator, UnsupervisedEvaluator, load_mask
    z, _ = _golden()
    name = 'binary_32_to_64'
    a, b = torch.from_numpy(z[f'{name}_a']), torch.from_numpy(z[f'{name}_b'])
    want = z[f'{name}_iou']
    ev = MeanEvaluator()
    for i in range(a.shape[0]):
        ev.log_iou(a[i].to(DEV), b[i].to(DEV)).log_intensity(a[i].to(DEV))
    np.testing.assert_allclose(ev.ious, want, rtol=0, atol=1e-7)
    assert abs(ev.mean_iou - want.mean()) < 1e-7 and abs(ev.ci95_miou - 1.96 * want.std() / np.sqrt(len(want))) < 1e-7
    assert abs(ev.mean_intensity - float(np.mean([a[i].mean() for i in range(a.shape[0])]))) < 1e-6 and len(ev) == a.shape[0]
    # every prediction of the group as a candidate for truth 0 (CPU tensors: moved to the device), plus one of another size
    import daam_amd
    singles = [daam_amd.compute_iou(a[i], b[0]) for i in range(a.shape[0])]
    big = torch.from_numpy(z['same_binary_64_a'][0])
    singles.append(daam_amd.compute_iou(big, b[0]))
    ev2 = MeanEvaluator().log_iou([a[i] for i in range(a.shape[0])] + [big], b[0])
    assert ev2.ious == [max(singles)]
    un = UnsupervisedEvaluator()
    for gt in range(2):
        for pred in range(2):
            un.log_iou(a[pred].to(DEV), b[gt].to(DEV), gt_idx=gt, pred_idx=pred)
        un.increment()
    m = np.asarray([[daam_amd.compute_iou(a[p], b[g]) for p in range(2)] for g in range(2)])
    assert abs(un.mean_iou - max(m[0, 0] + m[1, 1], m[0, 1] + m[1, 0]) / 2) < 1e-7 and len(un) == 2
    # a mask file written by the reference, scored against itself
    mask = load_mask(os.path.join(GOLDEN_DIR, 'experiment_ref', 'p7', 'cat.gt.png'))
    assert abs(MeanEvaluator().log_iou(mask, mask).mean_iou - 1.0) < 1e-6


def test_parsed_heat_maps_on_the_device():
    """GlobalHeatMap.parsed_heat_maps / dependency_relations (heatmap.py:125-142) with the word maps from ``daam_word_heat_map``,
    against what the unmodified reference produced over the same stand-in parse."""
    import daam_amd
    from daam_amd import utils
    from oracle import fake_diffusers as fd
    from oracle.make_golden import PARSED_PROMPT, fake_parse
    z = np.load(os.path.join(GOLDEN_DIR, 'experiment.npz'))
    utils.set_nlp(fake_parse)
    try:
        ghm = daam_amd.GlobalHeatMap(fd.FakeTokenizer(), PARSED_PROMPT, torch.from_numpy(z['parsed_maps_in']).to(DEV))
        parsed = list(ghm.parsed_heat_maps())
        assert [p.token.text for p in parsed] == json.loads(str(z['parsed_tokens']))
        np.testing.assert_allclose(torch.stack([p.word_heat_map.heatmap for p in parsed]).cpu().numpy(), z['parsed_maps'], rtol=0, atol=1e-6)
        rels = list(ghm.dependency_relations())
        assert [[r.head_text, r.dep_text, r.relation] for r in rels] == json.loads(str(z['relations']))
        np.testing.assert_allclose(torch.stack([r.head_heat_map.heatmap for r in rels]).cpu().numpy(), z['relation_head_maps'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(torch.stack([r.dep_heat_map.heatmap for r in rels]).cpu().numpy(), z['relation_dep_maps'], rtol=0, atol=1e-6)
    finally:
        utils.set_nlp(None)

"""Disk formats and bookkeeping either side of the extraction path, against a directory
WRITTEN BY the unmodified reference and the results of its own loader / evaluators (``tests/golden/experiment_ref/``,
``tests/golden/experiment.npz``; generator: ``oracle/make_golden.py experiment``).  CPU only: file IO, pickles, label tables,
evaluator arithmetic -- the IoU kernel itself is covered by ``test_gpu_evaluate.py``."""
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
REF_DIR = os.path.join(GOLDEN, 'experiment_ref', 'p7')


@pytest.fixture(scope='module')
def z():
    return np.load(os.path.join(GOLDEN, 'experiment.npz'))


def _j(z, key):
    return json.loads(str(z[key]))


def test_label_tables_equal_the_reference(z):
    from daam_amd import coco, experiment
    ref = _j(z, 'labels')
    assert coco.COCO80_LABELS == ref['coco80'] and coco.COCO80_INDICES == ref['indices']
    assert coco.COCOSTUFF27_LABELS == ref['stuff27'] and coco.UNUSED_LABELS == ref['unused']
    assert coco.COCO80_ONTOLOGY == ref['ontology'] and list(coco.COCO80_ONTOLOGY) == list(ref['ontology'])
    assert coco.COCO80_TO_27 == ref['to27']
    assert coco.build_word_list_coco80() == ref['word_list'] and list(coco.build_word_list_coco80()) == list(ref['word_list'])
    # the names the reference's experiment module exports are importable from ours
    for name in ('GenerationExperiment', 'COCO80_LABELS', 'COCOSTUFF27_LABELS', 'COCO80_INDICES', 'build_word_list_coco80'):
        assert hasattr(experiment, name) and name in experiment.__all__


def test_load_mask_reads_the_alpha_channel(z, tmp_path):
    from daam_amd.evaluate import load_mask
    got = load_mask(os.path.join(REF_DIR, 'cat.gt.png'))
    assert got.dtype == torch.float32 and got.device.type == 'cpu'
    np.testing.assert_array_equal(got.numpy(), z['load_mask_cat'])
    # any non-zero alpha counts; the colour channels do not
    import PIL.Image
    px = np.zeros((3, 4, 4), dtype=np.uint8)                   # [h, w, rgba]
    px[..., :3] = 255
    px[0, 1, 3], px[2, 2, 3] = 1, 255
    PIL.Image.fromarray(px).save(tmp_path / 'm.png')
    want = np.zeros((3, 4), dtype=np.float32)
    want[0, 1] = want[2, 2] = 1
    np.testing.assert_array_equal(load_mask(str(tmp_path / 'm.png')).numpy(), want)
    PIL.Image.fromarray(px[..., 0]).save(tmp_path / 'grey.png')
    with pytest.raises(ValueError):
        load_mask(str(tmp_path / 'grey.png'))


def test_reference_written_experiment_loads(z):
    """generation.pt pickled by the reference (class daam.experiment.GenerationExperiment) + its mask / annotation files, read
    by this package under every option set the reference's loader was run with."""
    from daam_amd import GenerationExperiment
    import daam_amd.experiment
    for tag, kw in _j(z, 'option_sets').items():
        exp = GenerationExperiment.load(REF_DIR, subtype='run', **kw)
        assert type(exp) is daam_amd.experiment.GenerationExperiment
        assert exp.prompt == 'a cat and a dog' and exp.seed == 5 and exp.id == 'p7' and exp.subtype == 'run'
        assert exp.annotations == {'split': 'val'} and str(exp.path) == REF_DIR
        assert tuple(exp.global_heat_map.shape) == (5, 8, 8) and exp.image.size == (16, 16) and not exp.nsfw()
        for kind, masks in (('truth', exp.truth_masks), ('pred', exp.prediction_masks)):
            assert sorted(masks) == _j(z, f'{tag}_{kind}_names'), (tag, kind)
            for name, m in masks.items():
                np.testing.assert_array_equal(m.numpy(), z[f'{tag}_{kind}_{name}'], err_msg=f'{tag} {kind} {name}')
    assert GenerationExperiment.contains_truth_mask(REF_DIR) and GenerationExperiment.contains_truth_mask(os.path.dirname(REF_DIR), 'p7')
    assert not GenerationExperiment.contains_truth_mask(os.path.join(REF_DIR, 'run'))
    assert GenerationExperiment.has_annotations(REF_DIR) and not GenerationExperiment.has_annotations(os.path.join(REF_DIR, 'run'))
    assert GenerationExperiment.has_experiment(REF_DIR, 'run') and GenerationExperiment.read_seed(REF_DIR) == 5
    both = GenerationExperiment.load(REF_DIR, all_subtypes=True)
    assert [e.subtype for e in both] == ['run']                  # one sub-directory holds a checkpoint


def _decoded(path):
    import PIL.Image
    return np.asarray(PIL.Image.open(path))


def test_saved_directory_matches_the_reference_written_one(tmp_path):
    """The same experiment saved by this package: the same files, the same decoded pixels / text in every one of them."""
    from daam_amd import GenerationExperiment
    from oracle.make_golden import experiment_inputs
    import PIL.Image
    image, maps, truth, pred, composite = experiment_inputs()
    exp = GenerationExperiment(PIL.Image.fromarray(image), maps, 'a cat and a dog', seed=5, id='p7', path=str(tmp_path),
                               truth_masks=truth, subtype='run')
    exp.annotate('split', 'val').save(heat_maps=False)
    for word, m in pred.items():
        exp.save_prediction_mask(m, word, 'daam')
    exp.save_prediction_mask(pred['cat'], 'cat', 'other')
    ours = tmp_path / 'p7'
    listing = lambda root: sorted(os.path.relpath(os.path.join(r, f), root) for r, _, fs in os.walk(root) for f in fs)   # noqa: E731
    assert listing(ours) == [f for f in listing(REF_DIR) if 'composite' not in f]
    for rel in listing(ours):
        if rel.endswith('.png'):
            np.testing.assert_array_equal(_decoded(ours / rel), _decoded(os.path.join(REF_DIR, rel)), err_msg=rel)
        elif rel.endswith(('.txt', '.json')):
            assert (ours / rel).read_text() == open(os.path.join(REF_DIR, rel)).read(), rel
    back = GenerationExperiment.load(ours, subtype='run')
    assert torch.equal(back.global_heat_map, maps) and sorted(back.truth_masks) == ['cat', 'dog', 'sky']
    assert sorted(back.prediction_masks) == ['cat', 'dog', 'person']
    np.testing.assert_array_equal(back.prediction_masks['dog'].numpy(), pred['Dog'].numpy())
    exp.clear_prediction_masks('daam')
    assert sorted(GenerationExperiment.load(ours, subtype='run').prediction_masks) == []
    assert sorted(GenerationExperiment.load(ours, subtype='run', pred_prefix='other').prediction_masks) == ['cat']


def test_masks_folded_onto_one_name_are_united():
    from daam_amd.experiment import _add_mask
    a = torch.tensor([[1., 0.], [0., 0.]])
    b = torch.tensor([[1., 1.], [0., 0.]])
    masks = {}
    _add_mask(masks, 'cat', a, simplify80=True)
    _add_mask(masks, 'dog', b, simplify80=True)
    _add_mask(masks, 'person', b, simplify80=True)
    assert sorted(masks) == ['animal', 'person']
    assert masks['animal'].tolist() == [[1., 1.], [0., 0.]]      # union, clamped
    masks = {}
    _add_mask(masks, 'cat', a)
    assert sorted(masks) == ['cat'] and masks['cat'] is a


def test_evaluators_match_the_reference(z, monkeypatch):
    """MeanEvaluator / UnsupervisedEvaluator bookkeeping (best candidate, means, 1.96-sigma half width, Hungarian matching):
    the script the reference's classes were driven with, IoUs from the same look-up table."""
    from daam_amd import evaluate as ev
    table = {(int(a), int(b)): v for a, b, v in _j(z, 'evaluator_table')}
    calls = []

    def lookup(a, b):
        assert a.shape == b.shape and a.dim() == 3
        calls.append(a.shape[0])
        return np.asarray([table[(int(a[i].flatten()[0]), int(b[i].flatten()[0]))] for i in range(a.shape[0])], dtype=np.float32)
    monkeypatch.setattr(ev, 'compute_iou_batch', lookup)
    t = lambda v: torch.full((2, 2), float(v))                   # noqa: E731
    mean_ev, unsup = ev.MeanEvaluator(), ev.UnsupervisedEvaluator()
    script = _j(z, 'evaluator_script')
    for step in script:
        cands = [t(c) for c in step['cands']]
        assert mean_ev.log_iou(cands, t(step['truth'])).log_intensity(t(step['intensity'])) is mean_ev
        unsup.log_iou(cands, t(step['truth']), gt_idx=step['gt_idx'], pred_idx=step['pred_idx'])
        unsup.increment()
    assert calls == [len(s['cands']) for s in script for _ in (0, 1)]      # every candidate list = ONE batched call
    want = z['mean_evaluator']
    np.testing.assert_allclose([mean_ev.mean_iou, mean_ev.ci95_miou, mean_ev.mean_intensity], want[:3], rtol=1e-6)
    assert len(mean_ev) == int(want[3])
    np.testing.assert_allclose(unsup.mean_iou, z['unsupervised_evaluator'][0], rtol=1e-6)
    assert len(unsup) == int(z['unsupervised_evaluator'][1])
    assert str(mean_ev) == str(z['mean_evaluator_str']) and str(unsup) == str(z['unsupervised_evaluator_str'])
    # a single tensor is a one-candidate list; candidates of different shapes are scored group by group
    calls.clear()
    table[(7, 3)] = 0.25
    table[(8, 3)] = 0.75
    assert ev._best_iou(t(7), t(3)) == 0.25 and calls == [1]
    monkeypatch.setattr(ev, 'compute_iou_batch', lambda a, b: np.asarray([table[(int(a[i].flatten()[0]), 3)] for i in range(a.shape[0])]))
    assert ev._best_iou([t(7), torch.full((3, 3), 8.0)], t(3)) == 0.75
    with pytest.raises(ValueError):
        ev._best_iou([], t(3))


def test_parsed_heat_maps_and_dependency_relations(z, monkeypatch):
    """GlobalHeatMap.parsed_heat_maps / dependency_relations (heatmap.py:125-142) over the stand-in parse the reference was run
    on: the same tokens kept / skipped, the same arcs in the same order.  The word-map arithmetic is a kernel call
    (``test_gpu_evaluate.py`` runs this on the device): on this CPU-only suite it is replaced by its definition."""
    from oracle import fake_diffusers as fd
    from oracle.make_golden import PARSED_PROMPT, fake_parse
    import daam_amd
    from daam_amd import engine, utils
    monkeypatch.setattr(engine, 'word_heat_map', lambda maps, idxs: maps[list(idxs)].mean(0))
    utils.set_nlp(fake_parse)
    try:
        ghm = daam_amd.GlobalHeatMap(fd.FakeTokenizer(), PARSED_PROMPT, torch.from_numpy(z['parsed_maps_in']))
        parsed = list(ghm.parsed_heat_maps())
        assert [p.token.text for p in parsed] == _j(z, 'parsed_tokens')
        assert all(isinstance(p, daam_amd.ParsedHeatMap) and p.word_heat_map.word == p.token.text for p in parsed)
        np.testing.assert_allclose(np.stack([p.word_heat_map.heatmap.numpy() for p in parsed]), z['parsed_maps'], rtol=0, atol=1e-6)
        rels = list(ghm.dependency_relations())
        assert [[r.head_text, r.dep_text, r.relation] for r in rels] == _j(z, 'relations')
        np.testing.assert_allclose(np.stack([r.head_heat_map.heatmap.numpy() for r in rels]), z['relation_head_maps'], rtol=0, atol=1e-6)
        np.testing.assert_allclose(np.stack([r.dep_heat_map.heatmap.numpy() for r in rels]), z['relation_dep_maps'], rtol=0, atol=1e-6)
    finally:
        utils.set_nlp(None)


def test_cached_nlp_without_spacy_says_so():
    from daam_amd import utils
    utils.set_nlp(None)
    try:
        import spacy  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match='spaCy'):
            utils.cached_nlp('a dog')
    seen = []
    utils.set_nlp(lambda prompt: seen.append(prompt) or prompt.split(), type='toy')
    try:
        assert utils.cached_nlp('a dog', 'toy') == ['a', 'dog'] and utils.cached_nlp('a dog', 'toy') == ['a', 'dog']
        assert seen == ['a dog']                                   # parsed once, then served from the cache
    finally:
        utils.set_nlp(None, type='toy')

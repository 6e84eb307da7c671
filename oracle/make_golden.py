"""Generate ``tests/golden/*.npz`` by EXECUTING THE UNMODIFIED REFERENCE (read-only import of
``/root/reference/daam``) on fake SD-v1.5 / SDXL topologies.  TEST INFRASTRUCTURE.

Runs only in the build container (``/root/reference`` does not exist on the GPU box); the
resulting fixtures are committed and are what ``tests/`` compare against at run time.

    python -m oracle.make_golden            # regenerate every case
    python -m oracle.make_golden sd15_f32   # one case

What is the reference and what is scaffolding:
  * reference code executed as-is: ``UNetCrossAttentionLocator.locate`` (hook.py:95-127),
    ``DiffusionHeatMapHooker`` / ``UNetCrossAttentionHooker.__call__`` / ``_unravel_attn``
    (trace.py), ``RawHeatMapCollection.update`` (heatmap.py:153-156),
    ``compute_global_heat_map`` (trace.py:83-132), ``GlobalHeatMap.compute_word_heat_map``
    (heatmap.py:121-123), ``WordHeatMap.expand_as`` (heatmap.py:77-93).
  * scaffolding (``oracle/fake_diffusers.py``): the diffusers ``Attention`` restatement, the
    UNet block structure, the synthetic hidden states.  Projections are identity so the
    hidden states ARE Q / K and the fixtures pin exact input bits.
  * fp16 cases: the per-step path runs literally in fp16 on CPU.  ``compute_global_heat_map``
    on a GPU runs under ``autocast(float32)``, which up-casts the bicubic input to fp32
    (SURVEY.md section 5); with no GPU here autocast is disabled, so for fp16 cases the
    accumulated maps are cast to fp32 before the reference's ``compute_global_heat_map`` is
    called -- the only deviation from "unmodified", stated in the fixture meta.
"""
from __future__ import annotations

import json
import os
import sys
import warnings

import numpy as np
import torch

from oracle import fake_diffusers as fd

# DAAM_GOLDEN_OUT: write somewhere else (tools/golden_host_check.py regenerates into a scratch directory and compares)
OUT_DIR = os.environ.get('DAAM_GOLDEN_OUT') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def host_info() -> dict:
    """The host a fixture was generated on.  The fp32 cases do not depend on it (bit for bit on every host tried); the literal fp16 /
    bf16 cases do in their last bit: another CPU sums the fp16 GEMM in another order, which flips the rounding of an fp16 logit in
    0.01-0.2 % of the elements (tests/golden/PROVENANCE.json has the comparison between hosts)."""
    cpu = 'unknown'
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    try:
        isa = torch.backends.cpu.get_cpu_capability()
    except Exception:                                            # noqa: BLE001
        isa = 'unknown'
    return dict(cpu=cpu, cpu_capability=isa, torch=torch.__version__, threads=torch.get_num_threads())

LONG_PROMPT = ' '.join(f'w{i}' for i in range(75))     # 75 tokens -> 77 rows

CASES = {
    # config C1 flavour: SD-v1.5 topology, fp32, CFG batch 2, 5 steps, prompt 'a dog'
    'sd15_f32': dict(kind='sd15', dtype='float32', batch=2, steps=5, prompt='a dog', seed=11,
                     unet=dict(dim_head=8)),
    # same topology, literal fp16 pipeline, more steps so fp16 running sums round
    'sd15_f16': dict(kind='sd15', dtype='float16', batch=2, steps=12, prompt='a dog', seed=12,
                     unet=dict(dim_head=16)),
    # literal bf16 pipeline (bf16 logits / probabilities / running sums)
    'sd15_bf16': dict(kind='sd15', dtype='bfloat16', batch=2, steps=12, prompt='a dog', seed=18,
                      unet=dict(dim_head=16), variants=['default', 'normalize', 'factor_hi']),
    # SDXL topology (60 layers -> capped transformer blocks, fewer heads), factors {1,2}
    'sdxl_f32': dict(kind='sdxl', dtype='float32', batch=2, steps=2, prompt='a photo of a monkey', seed=13,
                     unet=dict(dim_head=8, heads_scale=0.2, tblocks_cap=2)),
    'sdxl_f16': dict(kind='sdxl', dtype='float16', batch=2, steps=6, prompt=LONG_PROMPT, seed=14,
                     unet=dict(dim_head=16, heads_scale=0.2, tblocks_cap=1),
                     variants=['default', 'normalize']),
    # SDXL asked for 2048x2048: config.sample_size stays 128, latent 256 -> factors {0,1},
    # bicubic x0.5 (down-sample, no antialias) and x1
    'sdxl2048_f32': dict(kind='sdxl', dtype='float32', batch=2, steps=1, prompt='a dog', seed=15,
                         unet=dict(dim_head=8, heads_scale=0.1, tblocks_cap=1, latent_size=256)),
    # reference quirks: no CFG (batch 1 keeps heads H/2..H-1, trace.py:240) and
    # num_images_per_prompt=2 under CFG (batch 4 -> "heads" = 2H)
    'sd15_nocfg_f32': dict(kind='sd15', dtype='float32', batch=1, steps=2, prompt='a dog', seed=16,
                           unet=dict(dim_head=8)),
    'sd15_b4_f32': dict(kind='sd15', dtype='float32', batch=4, steps=2, prompt='a dog', seed=17,
                        unet=dict(dim_head=8, heads_scale=0.25)),
    # save_heads=True: the ONLY configuration in which the locator also returns the mid block (trace.py:34-35).
    # SDXL: the mid block (32x32 at 1024 px) is captured with factor 2; SD-v1.5: located, but its 8x8 maps hit the
    # ``factor != 8`` gate (trace.py:289).  Every processor call writes ``{gen_idx}.pt`` (trace.py:246-247); a second
    # trace with load_heads=True on a pipeline fed DIFFERENT hidden states replays those files (trace.py:281-282).
    'sdxl_heads_f32': dict(kind='sdxl', dtype='float32', batch=2, steps=2, prompt='a photo of a monkey', seed=19,
                           unet=dict(dim_head=8, heads_scale=0.2, tblocks_cap=1), heads=True,
                           variants=['default', 'normalize', 'layer']),
    'sd15_heads_f16': dict(kind='sd15', dtype='float16', batch=2, steps=3, prompt='a dog', seed=20,
                           unet=dict(dim_head=16), heads=True, variants=['default', 'factor_hi']),
    # Attention flags of other checkpoints (SD-2.1-768 sets upcast_attention): f32 logits straight into the softmax /
    # softmax on up-cast fp16 logits (diffusers get_attention_scores)
    'sd15_upcast_attn_f16': dict(kind='sd15', dtype='float16', batch=2, steps=4, prompt='a dog', seed=21,
                                 unet=dict(dim_head=16, upcast_attention=True), variants=['default', 'normalize']),
    'sd15_upcast_softmax_f16': dict(kind='sd15', dtype='float16', batch=2, steps=4, prompt='a dog', seed=22,
                                    unet=dict(dim_head=16, upcast_softmax=True), variants=['default']),
    # SD-v1.5 at its REAL layer shapes (mini=False: 8 heads x head_dim 40 / 80 / 160 = 320 / 640 / 1280 channels at 64 / 32 / 16
    # squared): the unmodified reference on the shapes whose taps run as three kernels side by side (BASELINE configs[1])
    'sd15_real_f16': dict(kind='sd15', dtype='float16', batch=2, steps=3, prompt='a dog', seed=23, mini=False,
                          unet=dict(), variants=['default', 'normalize', 'factor_hi', 'factor_lo', 'layer_head']),
    # SDXL at its real layer shapes (60 tapped layers: 5 / 10 / 20 heads x head_dim 64, 1100 keys; BASELINE configs[2], the headline)
    'sdxl_real_f16': dict(kind='sdxl', dtype='float16', batch=2, steps=2, prompt='a dog', seed=24, mini=False,
                          unet=dict(), variants=['default', 'normalize', 'factor_hi']),
}

SAMPLE_TOKENS = [0, 1, 2, 76]
OUT_SAMPLE_ROWS = 6          # rows of each processor output kept verbatim in the fixture


def input_checksums(pipe, steps):
    """Cheap fingerprint of the synthetic inputs so a test can prove it regenerated the
    same bits."""
    s = []
    order = pipe.unet.execution_order()
    for i, spec in enumerate(order):
        x = pipe.hidden_states(i, spec, steps - 1).double()
        c = pipe.context(i, spec).double()
        s.append([float(x.sum()), float((x * x).sum()), float(c.sum()), float((c * c).sum())])
    return np.asarray(s, dtype=np.float64)


def output_fingerprint(pipe):
    """What every cross-attention processor call returned in the last step (reference trace.py:304), execution
    order: ``[n, 2]`` float64 (sum, sum of squares) + the first rows of batch element 1 verbatim."""
    outs = pipe.last_outputs
    sums = np.asarray([[float(o.double().sum()), float((o.double() ** 2).sum())] for o in outs])
    rows = [o[-1, :OUT_SAMPLE_ROWS].float().numpy() for o in outs]
    return sums, rows


def heads_fingerprint(data_dir, n_files):
    """``{gen_idx}.pt`` files written by save_heads (trace.py:246-247): shape, dtype, (sum, sum of squares)."""
    shapes, stats, dtypes = [], [], []
    for i in range(n_files):
        t = torch.load(os.path.join(data_dir, f'{i}.pt'))
        shapes.append(list(t.shape))
        dtypes.append(str(t.dtype))
        stats.append([float(t.double().sum()), float((t.double() ** 2).sum())])
    return np.asarray(shapes, dtype=np.int64), np.asarray(stats), dtypes


def run_case(name, spec, daam):
    import tempfile
    dtype = getattr(torch, spec['dtype'])
    pipe = fd.make_pipe(spec['kind'], dtype=dtype, batch=spec['batch'], seed=spec['seed'],
                        mini=spec.get('mini', True), identity_proj=True, **spec['unet'])
    pipe.keep_outputs = True
    out = {}
    heads_dir = tempfile.mkdtemp(prefix='daam_heads_') if spec.get('heads') else None
    trace_kw = dict(save_heads=True, data_dir=heads_dir) if heads_dir else {}
    with daam.trace(pipe, **trace_kw) as tc:
        pipe(spec['prompt'], num_inference_steps=spec['steps'], callback=tc.time_callback)
        out['out_sums'], rows = output_fingerprint(pipe)
        for i, r in enumerate(rows):
            out[f'out_rows_{i}'] = r
        if heads_dir:
            n_files = tc._gen_idx
            out['heads_n_files'] = np.asarray(n_files)
            out['heads_shapes'], out['heads_stats'], hd = heads_fingerprint(heads_dir, n_files)
            out['heads_dtypes'] = np.asarray(json.dumps(hd))
        items = list(tc.all_heat_maps)
        keys = np.asarray([k for k, _ in items], dtype=np.int32)
        out['keys'] = keys
        out['key_sum'] = np.asarray([float(v.double().sum()) for _, v in items])
        out['key_sumsq'] = np.asarray([float((v.double() ** 2).sum()) for _, v in items])
        # every (key, token) plane: its sum and a position-weighted sum (a transposed or shifted plane changes the second)
        out['plane_sum'] = np.stack([v.double().sum((1, 2)).numpy() for _, v in items])
        out['plane_wsum'] = np.stack([(v.double() * torch.arange(1, v.shape[1] * v.shape[2] + 1, dtype=torch.float64)
                                       .view(1, v.shape[1], v.shape[2])).sum((1, 2)).numpy() for _, v in items])
        out['raw_dtype'] = np.asarray(str(items[0][1].dtype))
        # one sampled raw map per distinct resolution (first + last key of that factor)
        sample_ids = []
        for f in sorted(set(keys[:, 0].tolist())):
            ids = np.nonzero(keys[:, 0] == f)[0]
            sample_ids += [int(ids[0]), int(ids[-1])]
        out['raw_sample_ids'] = np.asarray(sample_ids, dtype=np.int32)
        for sid in sample_ids:
            out[f'raw_{sid}'] = items[sid][1][SAMPLE_TOKENS].float().numpy()
        if dtype in (torch.float16, torch.bfloat16):
            # emulate CUDA autocast(float32): upsample_bicubic2d is on the FP32 policy list
            hm = tc.all_heat_maps.ids_to_heatmaps
            for k in list(hm.keys()):
                hm[k] = hm[k].float()
        layers = sorted(set(keys[:, 1].tolist()))
        heads = sorted(set(keys[:, 2].tolist()))
        factors = sorted(set(keys[:, 0].tolist()))
        variants = {
            'default': {},
            'normalize': dict(normalize=True),
            'factor_hi': dict(factors=[factors[-1]]),
            'factor_lo': dict(factors=[factors[0]]),
            'head': dict(head_idx=heads[len(heads) // 2]),
            'layer': dict(layer_idx=layers[len(layers) // 2]),
            'layer_head': dict(layer_idx=layers[-1], head_idx=heads[0], normalize=True),
        }
        if 'variants' in spec:
            variants = {vn: variants[vn] for vn in spec['variants']}
        for vn, kw in variants.items():
            ghm = tc.compute_global_heat_map(**kw)
            out[f'global_{vn}'] = ghm.heat_maps.float().numpy()
        out['variants'] = np.asarray(json.dumps(variants))
        ghm = tc.compute_global_heat_map()
        # next-row f1: word heat map + expand_as (heatmap.py:77-93, 121-123)
        word = spec['prompt'].split()[-1]
        whm = ghm.compute_word_heat_map(word)
        out['word'] = np.asarray(word)
        out['word_map'] = whm.heatmap.float().numpy()

        class _Img:
            size = (128, 128)
        out['word_expand_128'] = whm.expand_as(_Img()).numpy()
        out['word_expand_128_abs'] = whm.expand_as(_Img(), absolute=True).numpy()
        out['layer_names'] = np.asarray(json.dumps(tc.layer_names))
        out['last_prompt'] = np.asarray(tc.last_prompt)
        out['last_image'] = np.asarray(str(tc.last_image))
        out['time_idx'] = np.asarray(tc.time_idx)
    if heads_dir:
        # replay: a pipeline with OTHER hidden states (seed + 100) under load_heads=True reads the saved probabilities
        # back, so its maps equal the ones above and its attention outputs are the saved probabilities x its own V
        pipe2 = fd.make_pipe(spec['kind'], dtype=dtype, batch=spec['batch'], seed=spec['seed'] + 100,
                             mini=spec.get('mini', True), identity_proj=True, **spec['unet'])
        pipe2.keep_outputs = True
        with daam.trace(pipe2, load_heads=True, data_dir=heads_dir) as tc2:
            pipe2(spec['prompt'], num_inference_steps=spec['steps'])
            hm = tc2.all_heat_maps.ids_to_heatmaps
            if dtype in (torch.float16, torch.bfloat16):
                for k in list(hm.keys()):
                    hm[k] = hm[k].float()
            out['global_replay'] = tc2.compute_global_heat_map().heat_maps.float().numpy()
            out['replay_out_sums'], rows = output_fingerprint(pipe2)
        import shutil
        shutil.rmtree(heads_dir, ignore_errors=True)
    out['input_checksums'] = input_checksums(pipe, spec['steps'])
    meta = dict(spec)
    meta['reference'] = 'castorini/daam v0.2.0, executed unmodified via oracle/fake_diffusers.py'
    meta['fp16_note'] = 'fp16 accumulators cast to fp32 before compute_global_heat_map (CUDA autocast emulation)'
    meta['torch'] = torch.__version__
    meta['host'] = host_info()
    out['meta'] = np.asarray(json.dumps(meta))
    path = os.path.join(OUT_DIR, f'{name}.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: {len(keys)} keys, factors {factors}, global {out["global_default"].shape}, '
          f'{os.path.getsize(path) / 1e6:.2f} MB')


def evaluate_inputs():
    """Seeded (prediction, truth) pairs for the reference's compute_iou / compute_ioa (evaluate.py:14-35): binary blobs
    upscaled x8 and x2 (every bicubic weight of a power-of-two scale is exact in fp32, so the `>= 1` threshold is
    well-defined), a soft prediction upscaled to a NON-square truth, and same-height pairs (no resize, a used as is)."""
    g = torch.Generator().manual_seed(77)

    def blobs(n, h, w, p):
        x = torch.rand(n, 1, h // 4, w // 4, generator=g)
        x = torch.nn.functional.interpolate(x, size=(h, w), mode='nearest')[:, 0]
        return (x > p).float()
    pairs = {
        'binary_64_to_512': (blobs(3, 64, 64, 0.5), blobs(3, 512, 512, 0.6)),
        'binary_32_to_64': (blobs(4, 32, 32, 0.4), blobs(4, 64, 64, 0.5)),
        'soft_48x64_to_96x160': (torch.rand(3, 48, 64, generator=g) * 2.0, blobs(3, 96, 160, 0.5)),
        'same_binary_64': (blobs(3, 64, 64, 0.5), blobs(3, 64, 64, 0.5)),
        'same_soft_64': (torch.rand(3, 64, 64, generator=g), blobs(3, 64, 64, 0.3)),
    }
    return pairs


def run_evaluate(daam):
    """tests/golden/evaluate.npz: the unmodified reference's compute_iou / compute_ioa on evaluate_inputs()."""
    ev = sys.modules['daam.evaluate']
    out = {}
    for name, (a, b) in evaluate_inputs().items():
        out[f'{name}_a'] = a.numpy()
        out[f'{name}_b'] = b.numpy()
        out[f'{name}_iou'] = np.asarray([ev.compute_iou(a[i].clone(), b[i].clone()) for i in range(a.shape[0])], dtype=np.float64)
        out[f'{name}_ioa'] = np.asarray([ev.compute_ioa(a[i].clone(), b[i].clone()) for i in range(a.shape[0])], dtype=np.float64)
    out['names'] = np.asarray(json.dumps(list(evaluate_inputs())))
    out['meta'] = np.asarray(json.dumps(dict(reference='castorini/daam v0.2.0 daam/evaluate.py, executed unmodified',
                                             torch=torch.__version__)))
    path = os.path.join(OUT_DIR, 'evaluate.npz')
    np.savez_compressed(path, **out)
    print(f'evaluate: {len(evaluate_inputs())} groups, {os.path.getsize(path) / 1e6:.2f} MB')


EXPERIMENT_DIR = os.path.join(OUT_DIR, 'experiment_ref')

PARSED_PROMPT = 'a fluffy cat chases the elephant'
# a stand-in dependency parse of PARSED_PROMPT: (text, relation, index of the head token).  'quickly' and '.' are tokens the
# parser reports that the tokenizer's tokens of the prompt do not contain (skipped by the reference, heatmap.py:130,141).
PARSED_TOKENS = [('a', 'det', 2), ('fluffy', 'amod', 2), ('cat', 'nsubj', 3), ('chases', 'ROOT', 3), ('the', 'det', 5),
                 ('elephant', 'dobj', 3), ('quickly', 'advmod', 3), ('.', 'punct', 6)]


class ParsedToken:
    def __init__(self, text, dep):
        self.text, self.dep_, self.head = text, dep, None


def fake_parse(prompt):
    """What a spaCy pipeline would return, as far as the reference looks at it: tokens with ``.text`` / ``.dep_`` / ``.head``."""
    assert prompt == PARSED_PROMPT
    tokens = [ParsedToken(text, dep) for text, dep, _ in PARSED_TOKENS]
    for token, (_, _, head) in zip(tokens, PARSED_TOKENS):
        token.head = tokens[head]
    return tokens


def experiment_inputs():
    """Seeded inputs of the experiment fixture: a 16 x 16 RGB image, a [5, 8, 8] global map, three ground-truth masks (one
    name in mixed case, two names that ``simplify80`` folds onto the same coarse name), three predicted masks and a
    composite index image."""
    g = torch.Generator().manual_seed(91)
    image = (torch.rand(16, 16, 3, generator=g) * 255).to(torch.uint8).numpy()
    maps = torch.rand(5, 8, 8, generator=g)

    def mask(p):
        return (torch.rand(16, 16, generator=g) > p).float()
    truth = {'Cat': mask(0.5), 'dog': mask(0.6), 'sky': mask(0.4)}
    pred = {'cat': mask(0.5), 'Dog': mask(0.5), 'person': mask(0.7)}
    composite = torch.randint(0, 4, (16, 16), generator=g).to(torch.uint8).numpy()
    return image, maps, truth, pred, composite


def run_experiment(daam):
    """tests/golden/experiment_ref/ + experiment.npz: a directory WRITTEN BY the unmodified reference
    (``GenerationExperiment.save`` / ``save_prediction_mask``, experiment.py:140-167,218-221) and what its own
    ``GenerationExperiment.load`` reads back from it under the option sets below; evaluator results on fixed numbers; the
    label tables.  Scaffolding: the composite index image is written here with PIL (the reference only reads it), and
    ``torch.load`` of a pickled class needs TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1 under torch >= 2.6."""
    import shutil
    import PIL.Image
    os.environ['TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD'] = '1'
    ex = sys.modules['daam.experiment']
    ev = sys.modules['daam.evaluate']
    image, maps, truth, pred, composite = experiment_inputs()
    shutil.rmtree(EXPERIMENT_DIR, ignore_errors=True)
    exp = ex.GenerationExperiment(PIL.Image.fromarray(image), maps, 'a cat and a dog', seed=5, id='p7', path=EXPERIMENT_DIR,
                                  truth_masks=truth, subtype='run')
    exp.annotate('split', 'val').save(heat_maps=False)
    for word, m in pred.items():
        exp.save_prediction_mask(m, word, 'daam')
    exp.save_prediction_mask(pred['cat'], 'cat', 'other')
    PIL.Image.fromarray(composite).save(os.path.join(EXPERIMENT_DIR, 'p7', 'run', 'composite.comp.pred.png'))   # its own prefix: '*.daam.pred.png' would match it
    out = {}
    option_sets = {
        'default': dict(),
        'simplify80': dict(simplify80=True),
        'composite': dict(pred_prefix='comp', composite=True, vocab=['floor', 'cat', 'dog', 'car']),
        'composite_simplify80': dict(pred_prefix='comp', composite=True, simplify80=True, vocab=['floor', 'cat', 'dog', 'car']),
        'composite_novocab': dict(pred_prefix='comp', composite=True),
        'composite_missing': dict(pred_prefix='nothere', composite=True),
        'other_prefix': dict(pred_prefix='other'),
    }
    for tag, kw in option_sets.items():
        back = ex.GenerationExperiment.load(os.path.join(EXPERIMENT_DIR, 'p7'), subtype='run', **kw)
        for kind, masks in (('truth', back.truth_masks), ('pred', back.prediction_masks)):
            out[f'{tag}_{kind}_names'] = np.asarray(json.dumps(sorted(masks)))
            for name, m in masks.items():
                out[f'{tag}_{kind}_{name}'] = m.numpy()
        assert back.annotations == {'split': 'val'} and back.prompt == 'a cat and a dog' and back.seed == 5
        assert torch.equal(back.global_heat_map, maps)
    out['option_sets'] = np.asarray(json.dumps(option_sets))
    out['load_mask_cat'] = ev.load_mask(os.path.join(EXPERIMENT_DIR, 'p7', 'cat.gt.png')).numpy()
    # evaluators: bookkeeping on IoUs supplied from outside (compute_iou swapped for a table look-up in the generator AND in
    # the test; the IoU arithmetic itself is pinned by evaluate.npz)
    table = {}

    def fake_iou(a, b):
        return table[(int(a.flatten()[0]), int(b.flatten()[0]))]
    real = ev.compute_iou
    ev.compute_iou = fake_iou
    try:
        rng = np.random.RandomState(5)
        t = lambda v: torch.full((2, 2), float(v))             # noqa: E731
        mean_ev, unsup = ev.MeanEvaluator(), ev.UnsupervisedEvaluator()
        script = []
        for i in range(12):
            cands = [int(c) for c in rng.randint(0, 50, size=rng.randint(1, 4))]
            truth_id = int(rng.randint(0, 5))
            for c in cands:
                table.setdefault((c, truth_id), float(rng.rand()))      # (a pair met again keeps its value)
            gt_idx, pred_idx = int(rng.randint(0, 4)), int(rng.randint(0, 5))
            script.append(dict(cands=cands, truth=truth_id, gt_idx=gt_idx, pred_idx=pred_idx, intensity=float(rng.rand())))
            mean_ev.log_iou([t(c) for c in cands], t(truth_id)).log_intensity(t(script[-1]['intensity']))
            unsup.log_iou([t(c) for c in cands], t(truth_id), gt_idx=gt_idx, pred_idx=pred_idx)
            unsup.increment()
        out['evaluator_script'] = np.asarray(json.dumps(script))
        out['evaluator_table'] = np.asarray(json.dumps([[a, b, v] for (a, b), v in table.items()]))
        out['mean_evaluator'] = np.asarray([mean_ev.mean_iou, mean_ev.ci95_miou, mean_ev.mean_intensity, len(mean_ev)], dtype=np.float64)
        out['mean_evaluator_str'] = np.asarray(str(mean_ev))
        out['unsupervised_evaluator'] = np.asarray([unsup.mean_iou, len(unsup)], dtype=np.float64)
        out['unsupervised_evaluator_str'] = np.asarray(str(unsup))
    finally:
        ev.compute_iou = real
    # parsed_heat_maps / dependency_relations (heatmap.py:125-142) over the stand-in parse (scaffolding: daam.heatmap.cached_nlp
    # replaced by fake_parse; the reference's loops, lookups and skips run as they are)
    hm = sys.modules['daam.heatmap']
    real_nlp = hm.cached_nlp
    hm.cached_nlp = fake_parse
    try:
        g = torch.Generator().manual_seed(92)
        gmaps = torch.rand(10, 8, 8, generator=g)
        ghm = hm.GlobalHeatMap(fd.FakeTokenizer(), PARSED_PROMPT, gmaps)
        parsed = list(ghm.parsed_heat_maps())
        rels = list(ghm.dependency_relations())
        out['parsed_maps_in'] = gmaps.numpy()
        out['parsed_tokens'] = np.asarray(json.dumps([p.token.text for p in parsed]))
        out['parsed_maps'] = np.stack([p.word_heat_map.heatmap.numpy() for p in parsed])
        out['relations'] = np.asarray(json.dumps([[r.head_text, r.dep_text, r.relation] for r in rels]))
        out['relation_head_maps'] = np.stack([r.head_heat_map.heatmap.numpy() for r in rels])
        out['relation_dep_maps'] = np.stack([r.dep_heat_map.heatmap.numpy() for r in rels])
    finally:
        hm.cached_nlp = real_nlp
    out['labels'] = np.asarray(json.dumps(dict(coco80=ex.COCO80_LABELS, indices=ex.COCO80_INDICES, stuff27=ex.COCOSTUFF27_LABELS,
                                               ontology=ex.COCO80_ONTOLOGY, to27=ex.COCO80_TO_27, unused=ex.UNUSED_LABELS,
                                               word_list=ex.build_word_list_coco80())))
    out['meta'] = np.asarray(json.dumps(dict(reference='castorini/daam v0.2.0 daam/experiment.py + daam/evaluate.py, executed unmodified',
                                             torch=torch.__version__)))
    path = os.path.join(OUT_DIR, 'experiment.npz')
    np.savez_compressed(path, **out)
    size = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(EXPERIMENT_DIR) for f in fs)
    print(f'experiment: {len(option_sets)} option sets, {os.path.getsize(path) / 1e3:.1f} kB + directory {size / 1e3:.1f} kB')


def main(argv):
    warnings.filterwarnings('ignore')
    torch.set_num_threads(os.cpu_count() or 1)
    daam, _ = fd.import_reference()
    os.makedirs(OUT_DIR, exist_ok=True)
    names = argv or list(CASES) + ['evaluate', 'experiment']
    for n in names:
        if n == 'evaluate':
            run_evaluate(daam)
        elif n == 'experiment':
            run_experiment(daam)
        else:
            run_case(n, CASES[n], daam)


if __name__ == '__main__':
    main(sys.argv[1:])

"""The ONE stdout line of bench.py (CPU): round 5's line had grown to 25 KB and the driver recorded nothing (BENCH_r05.parsed = null).
The line is now built by ``bench.headline_line`` from the full record; these tests feed it the recorded full records of earlier
rounds (profiles/r0*_bench_*.json -- data, produced by bench.py on an MI355X) and check size, strictness and content."""
import glob
import json
import math
import os

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r0[3-9]_bench_*.json')))
CONTRACT = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
            'config', 'roofline', 'cpu_baseline')


def _strict(line):
    def no_constant(name):
        raise ValueError(f'non-strict JSON constant {name}')
    return json.loads(line, parse_constant=no_constant)


@pytest.mark.parametrize('path', RECORDS, ids=[os.path.basename(p) for p in RECORDS])
def test_headline_line_of_recorded_runs(path):
    full = json.load(open(path))
    if 'roofline' not in full or 'metric' not in full:
        pytest.skip('not a bench line record')
    line = bench.headline_line(bench._finite(full))
    assert '\n' not in line and len(line.encode()) < bench.LINE_LIMIT == 4096, len(line)
    got = _strict(line)
    for k in CONTRACT:
        assert k in got, k
    assert got['value'] == full['value'] and got['n_gpus'] == full['n_gpus'] and got['steps'] == full['steps']
    ro = got['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'bytes_per_launch', 'ms_per_launch'):
        assert k in ro, k
    assert ro['bound'] == 'hbm' and ro['unit'] == 'GB/s' and abs(ro['frac'] - ro['achieved'] / ro['peak']) < 1e-3
    assert 'workload' in got['config'] and 'model' not in got['config']
    if full.get('cpu_baseline'):
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in got['cpu_baseline'], k
    # nothing nested beyond the three objects of the contract
    assert all(not isinstance(v, (dict, list)) for k, v in got.items() if k not in ('config', 'roofline', 'cpu_baseline'))
    assert all(not isinstance(v, (dict, list)) for k in ('config', 'roofline') for v in got[k].values())
    # one scalar per extra leg made it into the line
    for name, o in (full.get('other_configs') or {}).items():
        assert got[f'{name}_maps_per_s'] == o['value']
        assert got[f'{name}_tap_frac'] == o['roofline']['frac']


def test_headline_line_survives_bloat_and_nan():
    """Oversized strings are cut, extras are dropped from the end before the contract's fields, NaN never reaches the line."""
    full = json.load(open(RECORDS[-1]))
    full['config']['collective'] = 'x' * 5000
    full['roofline']['kernel'] = 'k' * 3000
    full['other_configs'] = {f'leg{i}': dict(value=float(i), roofline=dict(frac=0.5, ms_per_launch=1.0, traffic_over_algorithmic=1.0),
                                             roofline_finalize=dict(frac=0.4, ms_per_launch=0.06)) for i in range(200)}
    full['sustained_maps_per_s'] = float('nan')
    full['roofline']['traffic'] = float('inf')
    line = bench.headline_line(bench._finite(full))
    assert len(line.encode()) < bench.LINE_LIMIT
    got = _strict(line)
    assert got['roofline']['traffic'] is None and 'sustained_maps_per_s' not in got
    assert len(got['config']['collective']) <= 160 and 'leg0_maps_per_s' in got and 'leg199_maps_per_s' not in got
    for k in CONTRACT:
        assert k in got
    with pytest.raises(ValueError):
        bench.headline_line(full)                                    # without _finite: allow_nan=False refuses


def test_multi_rank_line():
    """The --gpus N form: config names the collective, its world size and library; still one short strict line."""
    full = json.load(open(RECORDS[-1]))
    full.update(n_gpus=8, steps=4, cpu_baseline=None)
    full['config'].update(parallelism='prompt-shard x8', generations_per_rank=4, collective_world_size=8, collective_library='RCCL 2.26.6, world_size 8',
                          baseline_config='BASELINE.json configs[3]: SDXL-base-1.0 1024x1024, 50 steps, batch=32 prompts sharded 8xMI355X (RCCL gather)',
                          collective='all_gather of the final [4, 77, 64, 64] fp32 maps per rank over nccl')
    for k in ('other_configs', 'integrated', 'attend', 'reference_eager_mi355x', 'speedup_vs_eager_mi355x'):
        full.pop(k, None)
    got = _strict(bench.headline_line(bench._finite(full)))
    assert got['n_gpus'] == 8 and got['cpu_baseline'] is None
    assert got['config']['collective_world_size'] == 8 and got['config']['collective_library'].startswith('RCCL')
    assert 'configs[3]' in got['config']['baseline_config']


def test_finite():
    assert bench._finite({'a': [1.0, float('nan'), {'b': float('-inf')}], 'c': 'x'}) == {'a': [1.0, None, {'b': None}], 'c': 'x'}
    assert math.isfinite(bench._finite(2.5))

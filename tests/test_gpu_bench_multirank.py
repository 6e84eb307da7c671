"""bench.py's multi-rank branch on the ONE-GPU box: ``python bench.py --gpus 2 --dist-backend gloo --shared-device`` re-executes
itself under ``torch.distributed.run`` with two ranks that both drive the HIP path on cuda:0 (RCCL refuses two ranks on one
device, so the three collectives -- barrier, all_gather of the final maps, MAX of the elapsed times -- go over gloo).  What
runs here is everything the driver's 8-GPU launch runs except the RCCL transport itself: the respawn, the process-group
initialisation, rank-seeded inputs, the timed all_gather inside the barriers, the MAX all-reduce, every rank checking its
slice of the gathered maps, rank 0 printing ONE JSON line.  Run with ``-m gpu`` on an MI355X."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ['--steps', '3', '--warmup', '1', '--no-baselines', '--no-integrated', '--no-other-configs']


def _bench(*extra, timeout=600, **env_extra):
    """Runs bench.py; returns the parsed headline line (what the driver records: one strict-JSON line < 4 KB on stdout) with the
    full record (the file the line names) under ``['_full']``."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        full_path = os.path.join(tmp, 'bench_full.json')
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0', BENCH_FULL_RECORD=full_path, **env_extra)
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *COMMON, *extra], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=timeout)
        assert p.returncode == 0, p.stderr[-3000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
        assert len(lines) == 1, p.stdout[-2000:]                # rank 0 prints ONE line, nobody else does
        assert len(lines[0].encode()) < 4096, len(lines[0])

        def no_constant(name):
            raise ValueError(f'non-strict JSON constant {name}')
        rec = json.loads(lines[0], parse_constant=no_constant)
        assert rec['full_record'] == full_path
        rec['_full'] = json.load(open(full_path))
    return rec


def test_bench_two_ranks_one_device():
    one = _bench()
    two = _bench('--gpus', '2', '--dist-backend', 'gloo', '--shared-device')
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2
    assert two['steps'] == 3 and two['scaling'] == 'weak' and two['metric'] == one['metric']
    assert 'ONE DEVICE' in two['config']['collective'] and 'gloo' in two['config']['collective']
    assert two['config']['parallelism'] == 'prompt-shard x2'
    assert one['config']['collective_world_size'] == 1 and one['config']['collective_library'] is None
    assert two['config']['collective_world_size'] == 2 and two['config']['collective_library'].startswith('gloo')
    # the single-rank line measured its HBM traffic in the run (two children under rocprofv3 --pmc)
    # (a box without the profiler / without PMC access: bench.py degrades to the committed counters and says why -- not a failure)
    import shutil
    full_ro = one['_full']['roofline']
    if shutil.which('rocprofv3') is None or full_ro.get('traffic_in_run_note'):
        assert one['roofline']['traffic_measured_in_run'] is False and full_ro.get('traffic_in_run_note'), full_ro
    else:
        assert one['roofline']['traffic_measured_in_run'] is True, one['roofline']
        assert 0.9 <= one['roofline']['traffic_over_algorithmic'] <= 1.3, one['roofline']
    # two ranks share one GPU: the aggregate rate is about the single-rank rate (never the 2x of two devices), minus the
    # host-staged gather of 2 x 3 maps inside the timed region
    # (lower bound: the two ranks' 3-generation regions are 7 ms each next to a gather staged through the host and a gloo barrier,
    # measured 0.35-0.9x; below 0.3x something serialises that should not)
    assert 0.3 * one["value"] <= two["value"] <= 1.5 * one["value"], (one['value'], two['value'])
    assert two['roofline']['launches_per_generation'] == 1 and two['roofline']['frac'] > 0.1
    # the single-rank line carries the sustained-state figure next to the (short) timed region; multi-rank lines do not
    assert one['_full']['sustained']['seconds'] >= 2.0 and one['sustained_maps_per_s'] > 0 and one['sustained_tap_ms'] > 0, one['_full'].get('sustained')
    assert 'sustained' not in two['_full'] and 'sustained_maps_per_s' not in two
    aff = two['_full']['config']['rank_cpu_affinity']
    assert aff is None or 'cores' in aff
    # the line is the contract's fields + roofline + cpu_baseline + scalars; the detail lives in the full record
    for rec in (one, two):
        assert all(not isinstance(v, (dict, list)) for k, v in rec.items() if k not in ('config', 'roofline', 'cpu_baseline', '_full'))
        assert rec['_full']['value'] == rec['value'] and 'roofline_issue' in rec['_full'] and 'roofline_issue' not in rec
    assert one['warmup'] == 1 and one['warmup_effective'] >= 20


def test_bench_two_ranks_without_the_launcher_module():
    """``BENCH_NO_TORCHRUN=1``: the ranks are started by hand (what bench.py does when ``torch.distributed.run`` cannot be imported):
    same env:// rendezvous, same line."""
    two = _bench('--gpus', '2', '--dist-backend', 'gloo', '--shared-device', BENCH_NO_TORCHRUN='1')
    assert two['n_gpus'] == 2 and two['config']['collective_world_size'] == 2 and two['value'] > 0


def test_bench_eight_ranks_one_device():
    """The driver's largest launch shape -- ``--gpus 8`` -- on the one-GPU box: eight ranks under torch.distributed.run, rank-seeded
    inputs (4 step sets = 1.6 GB per rank), the gather of 8 x 3 maps, the MAX reduce, every rank's slice check.  If the 8-rank path
    has a defect that two ranks do not show (port / rendezvous, gather layout, rank-major slicing), it fails here and not on the
    driver's node."""
    rec = _bench('--gpus', '8', '--dist-backend', 'gloo', '--shared-device', '--pool', '4', timeout=900)
    assert rec['n_gpus'] == 8 and rec['steps'] == 3 and rec['scaling'] == 'weak'
    assert rec['config']['parallelism'] == 'prompt-shard x8'
    assert rec['config']['collective_world_size'] == 8 and 'world_size 8' in rec['config']['collective_library']
    assert rec['value'] > 0 and rec['roofline']['launches_per_generation'] == 1


def test_bench_comm_over_rccl_single_rank():
    """bench.py's collective layer on the PRODUCTION backend (``nccl`` = RCCL), as far as one GPU allows: a one-rank process group --
    communicator initialisation with ``device_id``, ``all_gather_into_tensor`` of device maps, the MAX all-reduce, the library stamp
    of the JSON line.  (RCCL refuses two ranks on one device; more ranks run over gloo above and on the driver's 8-GPU node.)"""
    code = (
        "import os, sys, torch\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "torch.cuda.set_device(0)\n"
        "comm = bench.Comm('nccl', torch.device('cuda', 0))\n"
        "assert (comm.world, comm.rank) == (1, 0)\n"
        "mine = torch.arange(3 * 77 * 64 * 64, device='cuda:0', dtype=torch.float32).view(3, 77, 64, 64)\n"
        "comm.barrier()\n"
        "got = comm.all_gather(mine)\n"
        "assert got.shape == mine.shape and torch.equal(got, mine)\n"
        "assert comm.max(2.5) == 2.5\n"
        "lib = comm.library()\n"
        "assert lib.startswith('RCCL') and 'world_size 1' in lib, lib\n"
        "comm.close()\n"
        "print('RCCL_OK', lib)\n")
    import socket
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0 and 'RCCL_OK' in p.stdout, (p.stdout[-500:], p.stderr[-2000:])


def test_bench_refuses_more_gpus_than_visible():
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *COMMON, '--gpus', str(n)], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and 'refusing' in (p.stderr + p.stdout)

"""world_size-2 gloo test of the multi-GPU exchange (prompt sharding + one all_gather of final
maps).  CPU only: the collective layer is exercised, not the kernels."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from daam_amd.distributed import gather_heat_maps, shard_indices


def test_shard_indices_cover_everything_once():
    for n in (0, 1, 5, 8, 33):
        for world in (1, 2, 3, 8):
            got = sorted(i for r in range(world) for i in shard_indices(n, r, world))
            assert got == list(range(n))
            sizes = [len(shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_indices(4, 2, 2)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_items, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        mine = shard_indices(n_items, rank, world)
        # item i's "heat map" is a [3, 4, 4] tensor filled with i + plane/10
        local = torch.stack([torch.full((3, 4, 4), float(i)) + torch.arange(3).view(3, 1, 1) / 10 for i in mine]) \
            if mine else torch.zeros(0, 3, 4, 4)
        full = gather_heat_maps(local, n_items)
        torch.save(full, os.path.join(out_dir, f'r{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_items', [4, 5, 1])
def test_gather_world2_gloo(tmp_path, n_items):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    want = torch.stack([torch.full((3, 4, 4), float(i)) + torch.arange(3).view(3, 1, 1) / 10 for i in range(n_items)])
    for r in range(world):
        got = torch.load(os.path.join(str(tmp_path), f'r{r}.pt'))
        assert torch.equal(got, want)


@pytest.mark.parametrize('world,n_items', [(3, 7), (3, 2), (8, 33), (8, 5)])
def test_gather_ragged_tails_gloo(tmp_path, world, n_items):
    """More than two ranks, shards of unequal length (3 ranks x 7 items: 3 / 2 / 2; 3 x 2 and 8 x 5: ranks with NOTHING, whose
    contribution is all padding; 8 x 33: 5 / 4 x 7): every rank gets every item, in item order."""
    mp.spawn(_worker, args=(world, _free_port(), n_items, str(tmp_path)), nprocs=world, join=True)
    want = torch.stack([torch.full((3, 4, 4), float(i)) + torch.arange(3).view(3, 1, 1) / 10 for i in range(n_items)])
    for r in range(world):
        assert torch.equal(torch.load(os.path.join(str(tmp_path), f'r{r}.pt')), want)


def test_gather_single_process_passthrough():
    x = torch.randn(3, 2, 2)
    assert gather_heat_maps(x, 3) is x
    with pytest.raises(ValueError):
        gather_heat_maps(x, 4)


def _comm_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    comm = bench.Comm('gloo', torch.device('cpu'))
    try:
        assert (comm.world, comm.rank) == (world, rank)
        mine = torch.full((3, 2, 4, 4), float(rank)) + torch.arange(3).view(3, 1, 1, 1) / 10
        comm.barrier()
        got = comm.all_gather(mine)                        # rank-major: rank r's maps are rows [3r, 3r + 3)
        assert torch.equal(got[rank * 3:(rank + 1) * 3], mine)
        slowest = comm.max(1.0 + rank)
        assert f'world_size {world}' in comm.library()
        torch.save(dict(got=got, slowest=slowest), os.path.join(out_dir, f'c{rank}.pt'))
    finally:
        comm.close()


@pytest.mark.parametrize('world', [2, 8])
def test_bench_comm_gloo(tmp_path, world):
    """bench.py's collective layer (barrier, all_gather of the final maps, MAX of the elapsed times) with two and with eight ranks
    (the driver's largest launch)."""
    mp.spawn(_comm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    want = torch.cat([torch.full((3, 2, 4, 4), float(r)) + torch.arange(3).view(3, 1, 1, 1) / 10 for r in range(world)])
    for r in range(world):
        rec = torch.load(os.path.join(str(tmp_path), f'c{r}.pt'))
        assert torch.equal(rec['got'], want) and rec['slowest'] == float(world)

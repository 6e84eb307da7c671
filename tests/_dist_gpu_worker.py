"""Worker of tests/test_gpu_distributed.py: one rank of a world_size-N job in which EVERY rank drives the HIP path on
cuda:0 (the GPU box has one device) and ``daam_amd.distributed.trace_prompts`` gathers the maps of all prompts.

    python tests/_dist_gpu_worker.py <backend> <rank> <world> <port> <out_dir>
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    backend, rank, world, port, out_dir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    from conftest import golden_pipe, load_golden
    from daam_amd.distributed import trace_prompts
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    if backend == 'nccl':
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        z, meta = load_golden('sd15_f16')
        pipe = golden_pipe(meta, device='cuda:0')
        prompts = ['a dog', 'a photo of a monkey', 'a cat', 'two dogs', 'a monkey riding a bicycle']
        maps, rows = trace_prompts(pipe, prompts, num_inference_steps=3)
        torch.cuda.synchronize()
        torch.save(dict(maps=maps.cpu(), rows=rows), os.path.join(out_dir, f'r{rank}.pt'))
        print(json.dumps(dict(rank=rank, backend=backend, ok=True)), flush=True)
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

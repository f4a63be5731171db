"""Can two ranks of ONE RCCL communicator share this box's single GPU?  (probe for a real N = 2 exchange on a one-GPU box)
   python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29641 tools/gpu/rccl_same_gpu_probe.py"""
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
try:
    dist.init_process_group('nccl', device_id=torch.device('cuda:0'))
    t = torch.full((4,), float(rank + 1), device='cuda:0')
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print(f'rank {rank}: all_reduce over {world} ranks on one GPU -> {t.tolist()}', flush=True)
    peer = 1 - rank
    s, r = torch.full((8,), float(rank), device='cuda:0'), torch.empty(8, device='cuda:0')
    ops = [dist.P2POp(dist.isend, s, peer), dist.P2POp(dist.irecv, r, peer)]
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    torch.cuda.synchronize()
    print(f'rank {rank}: send/recv with rank {peer} -> {r[0].item()}', flush=True)
    dist.destroy_process_group()
except Exception as exc:  # noqa: BLE001
    print(f'rank {rank}: FAILED: {type(exc).__name__}: {str(exc)[:400]}', flush=True)
    sys.exit(1)

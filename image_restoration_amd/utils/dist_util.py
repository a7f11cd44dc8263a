"""Process-group set-up: one process per GPU, torch.distributed over RCCL (backend 'nccl' IS RCCL on ROCm).

Counterpart of basicsr/utils/dist_util.py (init_dist :10-18, _init_dist_pytorch :21-25, get_dist_info :60-71,
master_only :74-82).  The slurm launcher is not reproduced (scheduler plumbing, out of scope)."""
import functools
import os

import torch
import torch.distributed as dist


def init_dist(launcher='pytorch', backend='nccl', **kwargs):
    if launcher != 'pytorch':
        raise ValueError(f'Invalid launcher type: {launcher}')
    rank = int(os.environ['RANK'])
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
        num_gpus = torch.cuda.device_count()
        torch.cuda.set_device(rank % num_gpus)
    dist.init_process_group(backend=backend, **kwargs)


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def master_only(func):

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        rank, _ = get_dist_info()
        if rank == 0:
            return func(*args, **kwargs)

    return wrapper

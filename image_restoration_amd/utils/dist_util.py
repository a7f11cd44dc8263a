"""Process-group set-up: one process per GPU, ``torch.distributed`` over RCCL (backend name 'nccl' on ROCm).

Behaviour of basicsr/utils/dist_util.py for the ``pytorch`` launcher (init_dist :10-25, get_dist_info :60-71, master_only :74-82);
the slurm launcher is scheduler plumbing and out of scope.  The rendezvous address defaults to 127.0.0.1 (single node; container
host names may not resolve)."""
import functools
import os

import torch
import torch.distributed as dist


def init_dist(launcher='pytorch', backend='nccl', **kwargs):
    """Reads RANK (set by torch.distributed.run) and joins the default process group; with RCCL the rank's GPU is
    ``rank mod visible devices``."""
    if launcher != 'pytorch':
        raise ValueError(f'Invalid launcher type: {launcher}')
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if backend == 'nccl':
        torch.cuda.set_device(int(os.environ['RANK']) % torch.cuda.device_count())
    dist.init_process_group(backend=backend, **kwargs)


def get_dist_info():
    """(rank, world size); (0, 1) outside a process group."""
    active = dist.is_available() and dist.is_initialized()
    return (dist.get_rank(), dist.get_world_size()) if active else (0, 1)


def master_only(func):
    """The wrapped function runs on rank 0 and is a no-op (returning None) elsewhere."""

    @functools.wraps(func)
    def on_rank0(*args, **kwargs):
        if get_dist_info()[0] != 0:
            return None
        return func(*args, **kwargs)

    return on_rank0

"""One rank of the data-parallel equivalence test (tests/test_dp_gpu.py).  Launched by torch.distributed.run with two ranks that share
the box's GPU (gloo rendezvous; RCCL refuses two ranks on one device).  Builds the model exactly the way train.py does — options dict with
dist=True, process seeded ``manual_seed + rank`` (utils/options.py) so every randomly initialised tensor DIFFERS between the ranks
before the model's replica alignment — runs ``--iters`` optimize_parameters on this rank's half of a fixed global batch and writes what
it ended up with."""
import argparse
import os
import sys
from collections import OrderedDict as OD

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def options(model_type, rank, world, dist, bf16=False):
    opt = OD(name='dp', model_type=model_type, scale=4, num_gpu=1, manual_seed=7, is_train=True, dist=dist, rank=rank, world_size=world)
    opt['network_g'] = OD(type='RRDBNet', num_in_ch=3, num_out_ch=3, num_feat=16, num_block=1, num_grow_ch=8)
    if model_type != 'SRModel':
        opt['network_d'] = OD(type='VGGStyleDiscriminator128', num_in_ch=3, num_feat=8) if model_type != 'ESRGANModel_unet' else \
            OD(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=8, skip_connection=True)
    if bf16:
        opt['network_g']['compute_dtype'] = 'bf16'
    opt['path'] = OD(pretrain_network_g=None, strict_load_g=True, resume_state=None)
    tr = OD(ema_decay=0.9)
    tr['optim_g'] = OD(type='Adam', lr=1e-3, weight_decay=0, betas=[0.9, 0.99])
    tr['optim_d'] = OD(type='Adam', lr=1e-3, weight_decay=0, betas=[0.9, 0.99])
    tr['scheduler'] = OD(type='MultiStepLR', milestones=[100], gamma=0.5)
    tr['pixel_opt'] = OD(type='L1Loss', loss_weight=1.0, reduction='mean')
    if model_type != 'SRModel':
        tr['gan_opt'] = OD(type='GANLoss', gan_type='vanilla', real_label_val=1.0, fake_label_val=0.0, loss_weight=5e-3)
    opt['train'] = tr
    if model_type == 'ESRGANModel_unet':
        opt['model_type'] = 'ESRGANModel'
    return opt


def global_batch(it, n):
    from image_restoration_amd.utils import synth
    return (torch.from_numpy(synth.uniform_input(4000 + it, (n, 3, 32, 32))), torch.from_numpy(synth.uniform_input(5000 + it, (n, 3, 128, 128))))


def snapshot(model):
    out = {}
    for label, pack in model.packs.items():
        out[f'{label}_params'] = pack.adam.flat_p.detach().cpu().numpy().copy()
        if pack.shadow_arena is not None:
            out[f'{label}_shadow'] = pack.shadow_arena.detach().cpu().numpy().copy()
        for name, buf in pack.net.named_buffers():
            out[f'{label}_buf_{name}'] = buf.detach().cpu().numpy().copy()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', required=True)
    ap.add_argument('--out', required=True)
    ap.add_argument('--iters', type=int, default=3)
    ap.add_argument('--per_rank', type=int, default=2)
    args = ap.parse_args()
    from image_restoration_amd.models import build_model
    from image_restoration_amd.utils.dist_util import get_dist_info, init_dist
    from image_restoration_amd.utils.options import set_random_seed
    init_dist('pytorch', backend='gloo')
    rank, world = get_dist_info()
    opt = options(args.model, rank, world, True)
    set_random_seed(opt['manual_seed'] + rank)          # as parse_options does: ranks draw different initial weights
    probe = torch.rand(4).numpy()                       # evidence that the ranks' RNG streams differ
    set_random_seed(opt['manual_seed'] + rank)
    model = build_model(opt)
    arrays = {'probe': probe}
    arrays.update({f'it0_{k}': v for k, v in snapshot(model).items()})
    if rank == 0:
        torch.save({k: v.detach().cpu() for k, v in model.net_g.state_dict().items()}, os.path.join(args.out, 'g_init.pth'))
    logs = []
    for it in range(1, args.iters + 1):
        model.update_learning_rate(it, warmup_iter=-1)
        lq, gt = global_batch(it, args.per_rank * world)
        sl = slice(rank * args.per_rank, (rank + 1) * args.per_rank)
        model.feed_data({'lq': lq[sl], 'gt': gt[sl]})
        model.optimize_parameters(it)
        if it == 1:
            arrays['it1_g_grad'] = model.optimizer_g.flat_g.detach().cpu().numpy().copy()   # after the exchange: the rank SUM
        arrays.update({f'it{it}_{k}': v for k, v in snapshot(model).items()})
        logs.append([model.get_current_log()[k] for k in sorted(model.get_current_log())])
    model.refresh_buffers()   # what the next step would start from: rank 0's buffers everywhere (DDP broadcast_buffers)
    arrays.update({f'itR_{k}': v for k, v in snapshot(model).items()})
    arrays['logs'] = np.array(logs, np.float64)
    arrays['log_keys'] = np.array(sorted(model.get_current_log()))
    np.savez(os.path.join(args.out, f'rank{rank}.npz'), **arrays)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

"""One rank under torch.distributed.run with backend 'nccl' (= RCCL on ROCm): every collective of the path through the RCCL API on device
tensors — the gradient-arena all-reduce and loss reduce of a data-parallel ESRGAN step, the replica-alignment broadcast, the tiler's
gather.  A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so this is world size 1: the communicator
is real, the exchange degenerate; the two-rank semantics are covered by the gloo tests (tests/test_dp_gpu.py, test_harness_*)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    args = ap.parse_args()
    import torch.distributed as dist
    from dp_worker import global_batch, options
    from image_restoration_amd.models import build_model
    from image_restoration_amd.tiling import _gather_crops, plan_tiles, tiled_forward
    from image_restoration_amd.utils.dist_util import get_dist_info, init_dist
    init_dist('pytorch', backend='nccl')
    rank, world = get_dist_info()
    assert (rank, world) == (0, 1) and dist.get_backend() == 'nccl'
    dev = torch.device('cuda')
    out = {'backend': dist.get_backend()}
    # data-parallel ESRGAN steps: all-reduce of both gradient arenas + reduce of the loss vector, all on device tensors over RCCL
    torch.manual_seed(7)
    model = build_model(options('ESRGANModel', rank, world, True))
    for pack in model.packs.values():
        pack.align_replicas()            # the constructor broadcast (skipped by the model at world size 1): through RCCL here
    ref = build_model(options('ESRGANModel', 0, 1, False))
    ref.net_g.load_state_dict(model.net_g.state_dict())
    ref.net_d.load_state_dict(model.net_d.state_dict())
    for m in (ref.net_g, ref.net_d):
        if hasattr(m, 'invalidate_packed'):
            m.invalidate_packed()
    ref.model_ema(0)
    model.model_ema(0)
    for it in range(1, 3):
        lq, gt = global_batch(it, 2)
        for mdl in (model, ref):
            mdl.update_learning_rate(it, warmup_iter=-1)
            mdl.feed_data({'lq': lq, 'gt': gt})
            mdl.optimize_parameters(it)
    out['params_equal_single_process'] = bool(torch.equal(model.optimizer_g.flat_p, ref.optimizer_g.flat_p) and
                                              torch.equal(model.optimizer_d.flat_p, ref.optimizer_d.flat_p))
    out['log'] = {k: float(v) for k, v in model.get_current_log().items()}
    # the tiler's exchange step on device buffers
    img = torch.rand(1, 3, 40, 56, device=dev)
    whole = tiled_forward(model.net_g.eval(), img, tile=16, pad=4, scale=4)
    cells = list(enumerate(plan_tiles(40, 56, 16, 4)))
    crops = {i: whole[:, :, c[0][0] * 4:c[0][1] * 4, c[0][2] * 4:c[0][3] * 4].contiguous() for i, c in cells}
    got = _gather_crops(crops, cells, 1, 0, 0, None, 3, 4, torch.float32, dev)
    out['gather_equal'] = all(torch.equal(got[i], crops[i]) for i, _ in cells)
    json.dump(out, open(os.path.join(args.out, 'rccl_one_rank.json'), 'w'))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()

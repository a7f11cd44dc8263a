"""BASELINE configs[2] (C3) at its real size under pytest: training_config/train_rrdbnet_esrgan_x4_mi355x_bf16_unet.yml UNSHRUNK —
23-block nf=64 RRDBNet in bf16, UNetDiscriminatorSN nf=64 in bf16 on 512x512, batch 32 of 128x128 LR patches, L1 + relativistic GAN,
two fused Adam steps, EMA.  This is the only place where the launch-size-selected code paths meet in one step: 8-wave 16/32-row bf16
tiles, the strided convs' zero-tap skipping (s2_channels), the one-launch dense-block weight gradient, the 14 GB saved-activation arena
and the 2^31 plane-offset guards.

The reference has no bf16 and no UNet discriminator, so the checks are the ones the domain offers at this size:
finite losses, bit-reproducibility of a repeated run (every reduction on the path is ordered), and agreement of the iteration-1
losses with the fp32 HIP path on the SAME weights within the declared bf16 tolerance (2e-2 relative: the tolerance of
tests/test_bf16_gpu.py's first-iteration check against the reference's float64 trajectory)."""
import os

import numpy as np
import pytest
import torch

from image_restoration_amd.models import build_model
from image_restoration_amd.utils import synth
from image_restoration_amd.utils.options import parse, set_random_seed

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
YML = os.path.join(ROOT, 'training_config', 'train_rrdbnet_esrgan_x4_mi355x_bf16_unet.yml')
BATCH, LQ = 32, 128


def _model(dtype):
    opt = parse(YML, ROOT, is_train=True)
    opt.update(dist=False, rank=0, world_size=1, num_gpu=1)
    assert opt['network_g']['num_block'] == 23 and opt['network_g']['num_feat'] == 64 and opt['network_d']['num_feat'] == 64
    assert opt['datasets']['train']['batch_size_per_gpu'] == BATCH and opt['datasets']['train']['gt_size'] == 4 * LQ
    opt['network_g']['compute_dtype'] = dtype
    opt['network_d']['compute_dtype'] = dtype
    set_random_seed(0)      # random initialisation of G, D and the spectral-norm vectors
    return build_model(opt)


def _batch(it, dev):
    return {'lq': torch.from_numpy(synth.uniform_input(300 + it, (BATCH, 3, LQ, LQ))).to(dev),
            'gt': torch.from_numpy(synth.uniform_input(400 + it, (BATCH, 3, 4 * LQ, 4 * LQ))).to(dev)}


def _run(model, dev, iters):
    logs = []
    for it in range(1, iters + 1):
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_data(_batch(it, dev))
        model.optimize_parameters(it)
        logs.append(dict(model.get_current_log()))
    torch.cuda.synchronize()
    return logs


def test_c3_full_size_step_is_finite_reproducible_and_tracks_fp32(cuda):
    first = _model('bf16')
    assert first.net_g.compute_dtype == 'bf16' and sum(p.numel() for p in first.net_g.parameters()) == 16_697_987
    start_g = first.optimizer_g.flat_p.clone()
    start_d = first.net_d.state_dict()
    start_d = {k: v.clone() for k, v in start_d.items()}
    logs_a = _run(first, cuda, 2)
    keys = sorted(logs_a[0])
    assert keys == ['l_d_fake', 'l_d_real', 'l_g_gan', 'l_g_pix', 'out_d_fake', 'out_d_real']
    assert all(np.isfinite(v) for log in logs_a for v in log.values()), logs_a
    end_g, end_d = first.optimizer_g.flat_p.clone(), first.optimizer_d.flat_p.clone()
    assert float((end_g - start_g).abs().max()) > 0 and bool(torch.isfinite(end_g).all()) and bool(torch.isfinite(end_d).all())
    ema = first.gen.shadow_arena.clone()
    del first
    torch.cuda.empty_cache()

    second = _model('bf16')                      # same seed -> same start; the whole step is deterministic
    assert torch.equal(second.optimizer_g.flat_p, start_g)
    logs_b = _run(second, cuda, 2)
    assert logs_b == logs_a
    assert torch.equal(second.optimizer_g.flat_p, end_g) and torch.equal(second.optimizer_d.flat_p, end_d)
    assert torch.equal(second.gen.shadow_arena, ema)
    del second
    torch.cuda.empty_cache()

    full = _model('fp32')                        # the fp32 HIP path on the same weights, one iteration
    assert torch.equal(full.optimizer_g.flat_p, start_g)
    assert all(torch.equal(v, start_d[k]) for k, v in full.net_d.state_dict().items())
    log32 = _run(full, cuda, 1)[0]
    for k in keys:
        assert abs(logs_a[0][k] - log32[k]) <= 2e-2 * max(abs(log32[k]), 1e-2), (k, logs_a[0][k], log32[k])

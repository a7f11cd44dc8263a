#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REFERENCE's own modules in place.

Build-container only (needs /root/reference; see tools/ref_loader.py).  Weights and inputs
come from image_restoration_amd.utils.synth (numpy PCG64, seed-stable) and are loaded into
the reference modules with load_state_dict(strict=True); only inputs/outputs are stored.
Vector ids follow SURVEY.md §8c (G-a ... G-l).

    PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py [--only g_a,g_d]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

from image_restoration_amd.utils import synth  # noqa: E402
from ref_loader import load_reference  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')


def to_torch(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB  ' + ', '.join(f'{k}{tuple(np.asarray(v).shape)}' for k, v in arrays.items()))


def g_a(ref):
    """G-a / G-b: one ResidualDenseBlock(64,32) on [1,64,12,12]: out, x1..x4; grads of sum(out*R)."""
    blk = ref.ResidualDenseBlock(64, 32)
    blk.load_state_dict(to_torch(synth.rdb_state_dict(11, 64, 32)), strict=True)
    x = torch.from_numpy(synth.signed_input(12, (1, 64, 12, 12))).requires_grad_(True)
    R = torch.from_numpy(synth.signed_input(13, (1, 64, 12, 12)))
    # intermediates through the reference's own convs (forward :33-36)
    with torch.no_grad():
        x1 = blk.lrelu(blk.conv1(x))
        x2 = blk.lrelu(blk.conv2(torch.cat((x, x1), 1)))
        x3 = blk.lrelu(blk.conv3(torch.cat((x, x1, x2), 1)))
        x4 = blk.lrelu(blk.conv4(torch.cat((x, x1, x2, x3), 1)))
    out = blk(x)
    (out * R).sum().backward()
    grads = {f'grad_{n.replace(".", "_")}': p.grad.numpy() for n, p in blk.named_parameters()}
    save('g_a_rdb', x=x.detach().numpy(), R=R.numpy(), out=out.detach().numpy(), x1=x1.numpy(), x2=x2.numpy(),
         x3=x3.numpy(), x4=x4.numpy(), grad_x=x.grad.numpy(), **grads)


def g_c(ref):
    """G-c: one RRDB(64,32) fwd/bwd on [1,64,8,8]."""
    blk = ref.RRDB(64, 32)
    blk.load_state_dict(to_torch(synth.rrdb_state_dict(21, 64, 32)), strict=True)
    x = torch.from_numpy(synth.signed_input(22, (1, 64, 8, 8))).requires_grad_(True)
    R = torch.from_numpy(synth.signed_input(23, (1, 64, 8, 8)))
    out = blk(x)
    (out * R).sum().backward()
    gn = np.array([float(p.grad.norm()) for _, p in blk.named_parameters()], dtype=np.float64)
    save('g_c_rrdb', x=x.detach().numpy(), R=R.numpy(), out=out.detach().numpy(), grad_x=x.grad.numpy(),
         grad_norms=gn, grad_rdb3_conv5_weight=blk.rdb3.conv5.weight.grad.numpy(),
         grad_rdb1_conv1_weight=blk.rdb1.conv1.weight.grad.numpy(), grad_rdb2_conv3_bias=blk.rdb2.conv3.bias.grad.numpy())


def g_d(ref):
    """G-d: BASELINE config 1 exactly: RRDBNet(3,3,4,32,1,32) on one 64x64 crop (+ uint8 pre/post, a9)."""
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=32, num_block=1, num_grow_ch=32)
    net = ref.RRDBNet(**cfg).eval()
    net.load_state_dict(to_torch(synth.rrdbnet_state_dict(0, **cfg)), strict=True)
    x = torch.from_numpy(synth.uniform_input(1234, (1, 3, 64, 64)))
    with torch.no_grad():
        y = net(x)
    # uint8 BGR HWC crop -> img2tensor convention (img_util.py:9-35, /255 at inference.py:68) -> net ->
    # tensor2img convention (img_util.py:38-94 with min_max=(0,1): clamp, CHW->HWC, RGB->BGR, *255 round)
    img = np.random.default_rng(77).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    t = torch.from_numpy(np.ascontiguousarray(img[:, :, ::-1].transpose(2, 0, 1))).float().div(255.)[None]
    with torch.no_grad():
        o = net(t)
    o = o.squeeze(0).float().clamp_(0, 1)
    out_img = (o.numpy().transpose(1, 2, 0)[:, :, ::-1] * 255.0).round().astype(np.uint8)
    save('g_d_c1', x=x.numpy(), y=y.numpy(), img_u8=img, out_u8=out_img)


def g_e(ref):
    """G-e: 23-block nf=64 on [1,3,24,24]: fwd; per-parameter grad L2 norms + two full grads."""
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32)
    net = ref.RRDBNet(**cfg)
    net.load_state_dict(to_torch(synth.rrdbnet_state_dict(0, **cfg)), strict=True)
    x = torch.from_numpy(synth.uniform_input(1234, (1, 3, 24, 24))).requires_grad_(True)
    R = torch.from_numpy(synth.signed_input(5, (1, 3, 96, 96)))
    y = net(x)
    (y * R).sum().backward()
    gn = np.array([float(p.grad.double().norm()) for _, p in net.named_parameters()], dtype=np.float64)
    save('g_e_full23', x=x.detach().numpy(), R=R.numpy(), y=y.detach().numpy(), grad_x=x.grad.numpy(), grad_norms=gn,
         grad_conv_first_weight=net.conv_first.weight.grad.numpy(),
         grad_body22_rdb3_conv5_weight=net.body[22].rdb3.conv5.weight.grad.numpy(),
         grad_conv_last_bias=net.conv_last.bias.grad.numpy())


def g_f(ref):
    """G-f: head only ([1,64,6,6] through up1/up2/hr/last) — pins the nearest-gather indexing (:116-118)."""
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=0, num_grow_ch=32)
    net = ref.RRDBNet(**cfg).eval()
    net.load_state_dict(to_torch(synth.rrdbnet_state_dict(3, **cfg)), strict=True)
    feat = torch.from_numpy(synth.signed_input(31, (1, 64, 6, 6)))
    import torch.nn.functional as F
    with torch.no_grad():
        f = net.lrelu(net.conv_up1(F.interpolate(feat, scale_factor=2, mode='nearest')))
        f = net.lrelu(net.conv_up2(F.interpolate(f, scale_factor=2, mode='nearest')))
        out = net.conv_last(net.lrelu(net.conv_hr(f)))
        # non-multiple-of-32 spatial size through the whole (0-block) network
        x = torch.from_numpy(synth.uniform_input(32, (2, 3, 13, 37)))
        y = net(x)
    save('g_f_head', feat=feat.numpy(), out=out.numpy(), x=x.numpy(), y=y.numpy())


def g_l(ref):
    """G-l: scale=2 and scale=1 forward (pins pixel_unshuffle channel order, arch_util.py:200-201)."""
    arrays = {}
    for scale, hw in ((2, 16), (1, 16)):
        cfg = dict(num_in_ch=3, num_out_ch=3, scale=scale, num_feat=16, num_block=1, num_grow_ch=8)
        net = ref.RRDBNet(**cfg).eval()
        net.load_state_dict(to_torch(synth.rrdbnet_state_dict(40 + scale, **cfg)), strict=True)
        x = torch.from_numpy(synth.uniform_input(50 + scale, (2, 3, hw, hw)))
        with torch.no_grad():
            arrays[f'x_s{scale}'] = x.numpy()
            arrays[f'y_s{scale}'] = net(x).numpy()
        pu = ref.pixel_unshuffle(torch.arange(2 * 3 * 8 * 8, dtype=torch.float32).view(2, 3, 8, 8), scale if scale == 2 else 4)
        arrays[f'unshuffle_s{scale}'] = pu.numpy()
    save('g_l_scale', **arrays)


def g_g(ref):
    """G-g: VGGStyleDiscriminator128(3, 8): train-mode fwd/bwd on [4,3,128,128] incl. BN running-stat updates; eval fwd."""
    net = ref.VGGStyleDiscriminator128(3, 8)
    net.load_state_dict(to_torch(synth.vgg128_state_dict(61, 3, 8)), strict=True)
    net.train()
    x = torch.from_numpy(synth.uniform_input(62, (4, 3, 128, 128))).requires_grad_(True)
    R = torch.from_numpy(synth.signed_input(63, (4, 1)))
    out = net(x)
    (out * R).sum().backward()
    arrays = dict(x=x.detach().numpy(), R=R.numpy(), out_train=out.detach().numpy(), grad_x=x.grad.numpy())
    for n, p in net.named_parameters():
        arrays['grad_' + n.replace('.', '_')] = p.grad.numpy()
    for n, b in net.named_buffers():
        arrays['buf_' + n.replace('.', '_')] = b.detach().numpy()
    net.eval()
    with torch.no_grad():
        arrays['out_eval'] = net(x.detach()).numpy()
    save('g_g_vgg128', **arrays)


def g_n(ref):
    """G-n: VGGStyleDiscriminator256(3, 4): train-mode fwd/bwd on [2,3,256,256] (input regenerated from its seed), BN buffers,
    eval fwd; of dL/dx only the first image is stored."""
    net = ref.VGGStyleDiscriminator256(3, 4)
    net.load_state_dict(to_torch(synth.vgg128_state_dict(71, 3, 4, 256)), strict=True)
    net.train()
    x = torch.from_numpy(synth.uniform_input(72, (2, 3, 256, 256))).requires_grad_(True)
    R = torch.from_numpy(synth.signed_input(73, (2, 1)))
    out = net(x)
    (out * R).sum().backward()
    arrays = dict(R=R.numpy(), out_train=out.detach().numpy(), grad_x0=x.grad.numpy()[0])
    for n, p in net.named_parameters():
        arrays['grad_' + n.replace('.', '_')] = p.grad.numpy()
    for n, b in net.named_buffers():
        arrays['buf_' + n.replace('.', '_')] = b.detach().numpy()
    net.eval()
    with torch.no_grad():
        arrays['out_eval'] = net(x.detach()).numpy()
    save('g_n_vgg256', **arrays)


def g_h(ref):
    """G-h: L1Loss / GANLoss(vanilla) values + input grads on fixed tensors; the docstring vectors of weighted_loss
    (loss_util.py:78-85)."""
    arrays = {}
    pred = torch.from_numpy(synth.uniform_input(71, (2, 3, 16, 16))).requires_grad_(True)
    tgt = torch.from_numpy(synth.uniform_input(72, (2, 3, 16, 16)))
    l1 = ref.L1Loss(loss_weight=1e-2)
    loss = l1(pred, tgt)
    loss.backward()
    arrays.update(l1_pred=pred.detach().numpy(), l1_target=tgt.numpy(), l1_loss=loss.detach().numpy(), l1_grad=pred.grad.numpy())
    gan = ref.GANLoss('vanilla', loss_weight=5e-3)
    for name, shape in (('vec', (6, 1)), ('map', (2, 1, 8, 8))):
        a = torch.from_numpy(synth.signed_input(73, shape, 3.0)).requires_grad_(True)
        b = torch.from_numpy(synth.signed_input(74, shape, 3.0)).requires_grad_(True)
        arrays[f'gan_{name}_a'], arrays[f'gan_{name}_b'] = a.detach().numpy(), b.detach().numpy()
        for real in (True, False):
            for disc in (True, False):
                for rel in (False, True):
                    a.grad = b.grad = None
                    inp = a - torch.mean(b) if rel else a
                    l = gan(inp, real, is_disc=disc)
                    l.backward()
                    key = f'gan_{name}_real{int(real)}_disc{int(disc)}_rel{int(rel)}'
                    arrays[key + '_loss'] = l.detach().numpy()
                    arrays[key + '_ga'] = a.grad.numpy().copy()
                    arrays[key + '_gb'] = b.grad.numpy().copy() if rel else np.zeros(shape, np.float32)
    # docstring known-answer vectors of weighted_loss (loss_util.py:67-85)
    # (the 1-D weighted calls of that docstring trip the function's own size(1) assert, so they are run as [1,3])
    p = torch.Tensor([[0, 2, 3]]); t = torch.Tensor([[1, 1, 1]]); w = torch.Tensor([[1, 0, 1]])
    arrays['doc_mean'] = ref.l1_loss(p, t).numpy()
    arrays['doc_weighted_mean'] = ref.l1_loss(p, t, w).numpy()
    arrays['doc_none'] = ref.l1_loss(p, t, reduction='none').numpy()
    arrays['doc_weighted_sum'] = ref.l1_loss(p, t, w, reduction='sum').numpy()
    save('g_h_losses', **arrays)


def g_o(ref):
    """G-o: the other criteria of the reference on fixed tensors: MSELoss, CharbonnierLoss (mean / sum) and GANLoss with lsgan,
    wgan, wgan_softplus, hinge and soft-label vanilla — values and input gradients, plain and with the mean of a second
    tensor subtracted (the relativistic form of esrgan_model.py:40-41)."""
    arrays = {}
    pred0 = synth.uniform_input(81, (2, 3, 12, 20))
    tgt = torch.from_numpy(synth.uniform_input(82, (2, 3, 12, 20)))
    arrays.update(pix_pred=pred0, pix_target=tgt.numpy())
    for key, crit in (('mse_mean', ref.MSELoss(loss_weight=0.7)), ('mse_sum', ref.MSELoss(loss_weight=1.0, reduction='sum')),
                      ('charb_mean', ref.CharbonnierLoss(loss_weight=2.0, eps=1e-6)),
                      ('charb_sum', ref.CharbonnierLoss(loss_weight=1.0, reduction='sum', eps=1e-12))):
        pred = torch.from_numpy(pred0).requires_grad_(True)
        loss = crit(pred, tgt)
        loss.backward()
        arrays[key + '_loss'], arrays[key + '_grad'] = loss.detach().numpy(), pred.grad.numpy()
    a0, b0 = synth.signed_input(83, (3, 1, 6, 10), 2.5), synth.signed_input(84, (3, 1, 6, 10), 2.5)
    arrays.update(gan_a=a0, gan_b=b0)
    for kind, rl, fl in (('vanilla', 0.9, 0.1), ('lsgan', 1.0, 0.0), ('lsgan', 0.8, 0.2), ('wgan', 1.0, 0.0), ('wgan_softplus', 1.0, 0.0),
                         ('hinge', 1.0, 0.0)):
        gan = ref.GANLoss(kind, real_label_val=rl, fake_label_val=fl, loss_weight=0.3)
        for real in (True, False):
            for disc in (True, False):
                for rel in (False, True):
                    a, b = torch.from_numpy(a0).requires_grad_(True), torch.from_numpy(b0).requires_grad_(True)
                    l = gan(a - torch.mean(b) if rel else a, real, is_disc=disc)
                    l.backward()
                    key = f'{kind}_{rl}_real{int(real)}_disc{int(disc)}_rel{int(rel)}'
                    arrays[key + '_loss'], arrays[key + '_ga'] = l.detach().numpy(), a.grad.numpy().copy()
                    if rel:
                        arrays[key + '_gb'] = b.grad.numpy().copy()
    save('g_o_losses2', **arrays)


def g_p(ref):
    """G-p: paired_random_crop / mod_crop of the reference's transforms.py (plain numpy + Python's random): for seeds 0..15 the
    LQ / GT windows drawn from a coordinate-coded pair, single images and lists of two."""
    import random
    T = ref.transforms
    yy, xx = np.meshgrid(np.arange(20), np.arange(24), indexing='ij')
    lq = np.stack([yy, xx, (yy + xx) % 7], axis=2).astype(np.float32)
    gt = lq.repeat(4, 0).repeat(4, 1)
    arrays = {}
    for seed in range(16):
        random.seed(seed)
        g, l = T.paired_random_crop(gt, lq, 32, 4)
        arrays[f'lq_{seed}'], arrays[f'gt_corner_{seed}'] = l, g[::31, ::31].copy()
    random.seed(99)
    gs, ls = T.paired_random_crop([gt, gt + 1], [lq, lq + 1], 16, 4)
    arrays.update(list_lq0=ls[0], list_lq1=ls[1], list_gt1_corner=gs[1][::15, ::15].copy())
    arrays['mod_crop_shape'] = np.array(T.mod_crop(gt[:79, :93], 4).shape)
    save('g_p_crop', **arrays)


def g_q(ref):
    """G-q: calculate_psnr of the reference (psnr_ssim.py:8-46) on a fixed uint8 pair: crop_border 0 / 4, HWC / CHW, RGB and the
    y channel (BT.601 conversion of matlab_functions.bgr2ycbcr), and the identical-image case."""
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (40, 52, 3), dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-12, 13, a.shape), 0, 255).astype(np.uint8)
    arrays = dict(img1=a, img2=b)
    for cb in (0, 4):
        for y in (False, True):
            arrays[f'psnr_cb{cb}_y{int(y)}'] = np.float64(ref.calculate_psnr(a, b, cb, 'HWC', y))
            arrays[f'psnr_chw_cb{cb}_y{int(y)}'] = np.float64(ref.calculate_psnr(a.transpose(2, 0, 1), b.transpose(2, 0, 1), cb, 'CHW', y))
    arrays['psnr_same'] = np.float64(ref.calculate_psnr(a, a, 0))
    save('g_q_psnr', **arrays)


def g_r(ref):
    """G-r: the reference's options.parse (options.py:37-95) and dict2str on this repository's own option files, train / test
    mode and --debug, with a fixed root path; stored as JSON text."""
    import json
    files = [('options/train/ESRGAN/train_ESRGAN_x4_synthetic.yml', True), ('options/train/ESRGAN/train_RRDBNet_PSNR_x4_folders.yml', True),
             ('training_config/train_rrdbnet_esrgan_x4_mi355x_bf16_unet.yml', True), ('options/test/ESRGAN/test_ESRGAN_x4.yml', False),
             ('options/test/ESRGAN/test_ESRGAN_x4_woGT.yml', False)]
    arrays = {}
    for i, (rel, is_train) in enumerate(files):
        for debug in (False, True):
            opt = ref.options.parse(os.path.join(ROOT, rel), '/srv/run', is_train=is_train, debug=debug)
            arrays[f'f{i}_d{int(debug)}_json'] = np.array(json.dumps(opt, sort_keys=True))
            if not debug:
                arrays[f'f{i}_str'] = np.array(ref.options.dict2str(opt))
    arrays['files'] = np.array(json.dumps(files))
    save('g_r_options', **arrays)


PAIR_NAMES = ['0801.png', '0802.png', 'a10.png', 'a9.png', 'b.PNG']


def g_s(ref):
    """G-s: paired_paths_from_folder / paired_paths_from_meta_info_file of the reference's data_util.py on a fixed set of file
    names (empty files in a temporary tree), with a filename template; stored relative to the tree."""
    import json
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        gt_dir, lq_dir = os.path.join(tmp, 'gt'), os.path.join(tmp, 'lq')
        os.makedirs(gt_dir), os.makedirs(lq_dir)
        for n in PAIR_NAMES:
            base, ext = os.path.splitext(n)
            open(os.path.join(gt_dir, n), 'w').close()
            open(os.path.join(lq_dir, f'{base}x4{ext}'), 'w').close()
        rel = lambda paths: [{k: os.path.relpath(v, tmp) for k, v in d.items()} for d in paths]
        a = ref.data_util.paired_paths_from_folder([lq_dir, gt_dir], ['lq', 'gt'], '{}x4')
        meta = os.path.join(tmp, 'meta.txt')
        with open(meta, 'w') as f:
            f.write('a9.png (480,480,3)\n0802.png (480,480,3)\n')
        b = ref.data_util.paired_paths_from_meta_info_file([lq_dir, gt_dir], ['lq', 'gt'], meta, '{}x4')
    save('g_s_paths', folder=np.array(json.dumps(rel(a))), meta=np.array(json.dumps(rel(b))), names=np.array(json.dumps(PAIR_NAMES)))


def _esrgan_opt(model_type, ema):
    from collections import OrderedDict as OD
    opt = OD(name='golden', model_type=model_type, scale=4, num_gpu=0, manual_seed=0, is_train=True, dist=False, rank=0,
             world_size=1)
    opt['network_g'] = OD(type='RRDBNet', num_in_ch=3, num_out_ch=3, num_feat=16, num_block=1, num_grow_ch=8)
    opt['network_d'] = OD(type='VGGStyleDiscriminator128', num_in_ch=3, num_feat=8)
    opt['path'] = OD(pretrain_network_g=None, strict_load_g=True, resume_state=None)
    tr = OD(ema_decay=ema)
    tr['optim_g'] = OD(type='Adam', lr=1e-3, weight_decay=0, betas=[0.9, 0.99])
    tr['optim_d'] = OD(type='Adam', lr=1e-3, weight_decay=0, betas=[0.9, 0.99])
    tr['scheduler'] = OD(type='MultiStepLR', milestones=[2, 3], gamma=0.5)
    tr['total_iter'] = 4
    tr['warmup_iter'] = -1
    tr['pixel_opt'] = OD(type='L1Loss', loss_weight=1e-2, reduction='mean')
    tr['gan_opt'] = OD(type='GANLoss', gan_type='vanilla', real_label_val=1.0, fake_label_val=0.0, loss_weight=5e-3)
    tr['net_d_iters'] = 1
    tr['net_d_init_iters'] = 0
    opt['train'] = tr
    return opt


def _checksums(net):
    return np.array([[float(p.detach().double().sum()), float(p.detach().double().norm())] for _, p in net.named_parameters()])


def g_i(ref):
    """G-i: full optimize_parameters x3 iterations for ESRGANModel (relativistic), SRGANModel and SRModel on tiny
    nets (G nb=1/nf=16/gc=8, D VGG128 nf=8, batch 4, 32^2 -> 128^2): per-iteration losses, learning rates,
    post-step parameter checksums, Adam moment checksums, EMA checksums, BN running stats."""
    arrays = {}
    cfg_g = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=16, num_block=1, num_grow_ch=8)
    # Each model runs twice: in float32 (what the reference computes) and in float64 (ground truth of the same
    # recipe).  Tests bound |hip - f64| by a multiple of |ref_f32 - f64| instead of guessing tolerances for a
    # trajectory that is chaotic under Adam.
    runs = [(mt, cls, dt) for mt, cls in (('ESRGANModel', ref.ESRGANModel), ('SRGANModel', ref.SRGANModel),
                                          ('SRModel', ref.SRModel)) for dt in (torch.float32, torch.float64)]
    for mt0, cls, dt in runs:
        mt = mt0 if dt == torch.float32 else mt0 + '64'
        opt = _esrgan_opt(mt0, 0.9)
        if mt0 == 'SRModel':
            opt['train'].pop('gan_opt'); opt.pop('network_d'); opt['train'].pop('optim_d')
        model = cls(opt)
        for net in (model.net_g, getattr(model, 'net_g_ema', None), getattr(model, 'net_d', None)):
            if net is not None:
                net.to(dt)
        model.net_g.load_state_dict(to_torch(synth.rrdbnet_state_dict(81, **cfg_g)), strict=True)
        if hasattr(model, 'net_g_ema'):
            model.model_ema(0)
        if hasattr(model, 'net_d'):
            model.net_d.load_state_dict(to_torch(synth.vgg128_state_dict(82, 3, 8)), strict=True)
        logs = []
        lrs = []
        for it in range(1, 4):
            model.update_learning_rate(it, warmup_iter=-1)
            lrs.append(model.get_current_learning_rate()[0])
            lq = torch.from_numpy(synth.uniform_input(900 + it, (4, 3, 32, 32))).to(dt)
            gt = torch.from_numpy(synth.uniform_input(950 + it, (4, 3, 128, 128))).to(dt)
            model.feed_data({'lq': lq, 'gt': gt})
            model.optimize_parameters(it)
            log = model.get_current_log()
            logs.append([log[k] for k in sorted(log)])
            arrays[f'{mt}_g_checksum_it{it}'] = _checksums(model.net_g)
            if hasattr(model, 'net_d'):
                arrays[f'{mt}_d_checksum_it{it}'] = _checksums(model.net_d)
        arrays[f'{mt}_log_keys'] = np.array(sorted(log))
        arrays[f'{mt}_logs'] = np.array(logs, dtype=np.float64)
        arrays[f'{mt}_lrs'] = np.array(lrs, dtype=np.float64)
        arrays[f'{mt}_ema_checksum'] = _checksums(model.net_g_ema)
        st = model.optimizer_g.state_dict()['state']
        arrays[f'{mt}_adam_g_exp_avg'] = np.array([float(st[i]['exp_avg'].double().norm()) for i in sorted(st)])
        arrays[f'{mt}_adam_g_exp_avg_sq'] = np.array([float(st[i]['exp_avg_sq'].double().norm()) for i in sorted(st)])
        if hasattr(model, 'net_d'):
            arrays[f'{mt}_d_bn4_1_running_mean'] = model.net_d.bn4_1.running_mean.numpy().copy()
            arrays[f'{mt}_d_bn0_1_running_var'] = model.net_d.bn0_1.running_var.numpy().copy()
            arrays[f'{mt}_d_nbt'] = model.net_d.bn0_1.num_batches_tracked.numpy().copy()
        # final full tensors of two parameters
        arrays[f'{mt}_g_conv_last_weight'] = model.net_g.conv_last.weight.detach().double().numpy().copy()
        arrays[f'{mt}_g_rdb1_conv1_weight'] = model.net_g.body[0].rdb1.conv1.weight.detach().double().numpy().copy()
    save('g_i_steps', **arrays)


def g_j(ref):
    """G-j: LR sequences of MultiStepRestartLR / CosineAnnealingRestartLR for yml-style settings."""
    arrays = {}
    def run(sched_cls, n, **kw):
        p = torch.nn.Parameter(torch.zeros(1))
        o = torch.optim.Adam([p], lr=2e-4)
        s = sched_cls(o, **kw)
        out = []
        for it in range(1, n + 1):
            if it > 1:
                o.step(); s.step()
            out.append(o.param_groups[0]['lr'])
        return np.array(out, dtype=np.float64)
    arrays['multistep'] = run(ref.lr_scheduler.MultiStepRestartLR, 60, milestones=[10, 20, 40, 50], gamma=0.5)
    arrays['multistep_restart'] = run(ref.lr_scheduler.MultiStepRestartLR, 60, milestones=[10, 20, 35, 50], gamma=0.5,
                                      restarts=[0, 30], restart_weights=[1, 0.5])
    arrays['cosine'] = run(ref.lr_scheduler.CosineAnnealingRestartLR, 40, periods=[10, 10, 10, 10],
                           restart_weights=[1, 0.5, 0.5, 0.5], eta_min=1e-7)
    save('g_j_lr', **arrays)


def g_k(ref):
    """G-k: tiled inference of a 40x56 frame, tile 16, pad 4 (the build's paste rule, image_restoration_amd/tiling.py)
    with the REFERENCE RRDBNet.forward applied to every padded cell."""
    from image_restoration_amd.tiling import plan_tiles
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=32, num_block=2, num_grow_ch=16)
    net = ref.RRDBNet(**cfg).eval()
    net.load_state_dict(to_torch(synth.rrdbnet_state_dict(91, **cfg)), strict=True)
    img = torch.from_numpy(synth.uniform_input(92, (1, 3, 40, 56)))
    out = torch.zeros((1, 3, 160, 224))
    with torch.no_grad():
        for (y0, y1, x0, x1), (py0, py1, px0, px1) in plan_tiles(40, 56, 16, 4):
            sr = net(img[:, :, py0:py1, px0:px1])
            oy, ox = (y0 - py0) * 4, (x0 - px0) * 4
            out[:, :, y0 * 4:y1 * 4, x0 * 4:x1 * 4] = sr[:, :, oy:oy + (y1 - y0) * 4, ox:ox + (x1 - x0) * 4]
    save('g_k_tiled', img=img.numpy(), out=out.numpy())


def g_m(ref):
    """G-m: EnlargedSampler index streams (data_sampler.py:29-42) for 2 ranks, 3 epochs."""
    class DS:
        def __len__(self):
            return 10
    arrays = {}
    for rank in (0, 1):
        s = ref.data_sampler.EnlargedSampler(DS(), 2, rank, ratio=3)
        for epoch in (0, 1, 5):
            s.set_epoch(epoch)
            arrays[f'r{rank}_e{epoch}'] = np.array(list(iter(s)), dtype=np.int64)
        arrays[f'r{rank}_len'] = np.array(len(s))
    save('g_m_sampler', **arrays)


ALL = {'g_m': g_m, 'g_k': g_k, 'g_g': g_g, 'g_n': g_n, 'g_o': g_o, 'g_p': g_p, 'g_q': g_q, 'g_r': g_r, 'g_s': g_s, 'g_h': g_h, 'g_i': g_i, 'g_j': g_j, 'g_a': g_a, 'g_c': g_c, 'g_d': g_d, 'g_e': g_e, 'g_f': g_f, 'g_l': g_l}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = load_reference()
    names = [n for n in args.only.split(',') if n] or list(ALL)
    for n in names:
        ALL[n](ref)


if __name__ == '__main__':
    main()

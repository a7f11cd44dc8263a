"""UNetDiscriminatorSN on the HIP path against the oracle (oracle/unet_discriminator_ref.py).

**Parity unpinned by the reference**: the mounted reference has no UNetDiscriminatorSN; the oracle
restates the published architecture with torch.nn.utils.spectral_norm.  fp32 tolerances: forward 1e-4 relative,
gradients 5e-4 relative to each tensor's max (summation order)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import image_restoration_amd as ira
from image_restoration_amd import hip_autograd as A
from image_restoration_amd import hip_ops as H
from image_restoration_amd.utils import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def test_bilinear2x_forward_backward(cuda):
    for shape in [(2, 16, 5, 7), (1, 8, 1, 1), (1, 24, 8, 3)]:
        x = torch.from_numpy(synth.signed_input(1, shape)).requires_grad_(True)
        y = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False)
        g = torch.from_numpy(synth.signed_input(2, tuple(y.shape)))
        y.backward(g)
        xc = H.nchw_to_cb8(x.detach().to(cuda)).buf.requires_grad_(True)
        yc = A.Bilinear2xFn.apply(xc)
        assert _rel(H.cb8_to_nchw(H.CB8(yc.detach()), shape[1]), y) < 1e-6
        yc.backward(H.nchw_to_cb8(g.to(cuda)).buf)
        assert _rel(H.cb8_to_nchw(H.CB8(xc.grad), shape[1]), x.grad) < 1e-6


def test_spectral_norm_matches_torch(cuda):
    torch.manual_seed(3)
    conv = torch.nn.utils.spectral_norm(torch.nn.Conv2d(12, 20, 4, 2, 1, bias=False))
    w0, u0, v0 = conv.weight_orig.detach().clone(), conv.weight_u.clone(), conv.weight_v.clone()
    conv.train()
    x = torch.rand(2, 12, 8, 8)
    (conv(x) ** 2).sum().backward()           # one power iteration happened inside
    w_ref, g_ref = conv.weight.detach().clone(), conv.weight_orig.grad.clone()
    wo = w0.to(cuda).requires_grad_(True)
    u, v = u0.to(cuda), v0.to(cuda)
    w_sn = A.SpectralNormFn.apply(wo, u, v, True, 1e-12)
    assert _rel(w_sn, w_ref) < 1e-5 and _rel(u, conv.weight_u) < 1e-5 and _rel(v, conv.weight_v) < 1e-5
    (F.conv2d(x.to(cuda).cpu(), w_sn.cpu(), None, 2, 1) ** 2).sum().backward()  # conv on CPU: only the SN op is under test
    assert _rel(wo.grad, g_ref) < 1e-4
    # eval: stored u, v, no update
    conv.eval()
    with torch.no_grad():
        conv(x)
        w_eval = A.SpectralNormFn.apply(wo.detach(), u.clone(), v.clone(), False, 1e-12)
    assert _rel(w_eval, conv.weight) < 1e-5


def test_unet_discriminator_vs_oracle(cuda):
    from oracle.unet_discriminator_ref import UNetDiscriminatorSNRef
    torch.manual_seed(11)
    ref = UNetDiscriminatorSNRef(3, 16).train()
    net = ira.build_network(dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=16))
    assert list(net.state_dict().keys()) == list(ref.state_dict().keys())
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.to(cuda).train()
    x = torch.from_numpy(synth.uniform_input(5, (2, 3, 32, 48)))
    R = torch.from_numpy(synth.signed_input(6, (2, 1, 32, 48)))
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * R).sum().backward()
    xg = x.to(cuda).requires_grad_(True)
    y = net(xg)
    assert tuple(y.shape) == (2, 1, 32, 48) and _rel(y, yr) < 1e-4
    (y * R.to(cuda)).sum().backward()
    assert _rel(xg.grad, xr.grad) < 5e-4
    rp = dict(ref.named_parameters())
    for n, p in net.named_parameters():
        assert _rel(p.grad, rp[n].grad) < 5e-4, n
    rb = dict(ref.named_buffers())
    for n, b in net.named_buffers():
        assert _rel(b, rb[n]) < 1e-5, n
    ref.eval(), net.eval()
    with torch.no_grad():
        assert _rel(net(x.to(cuda)), ref(x)) < 1e-4
    with pytest.raises(AssertionError):
        net(torch.zeros(1, 3, 12, 16, device=cuda))


def test_esrgan_step_with_unet_discriminator_on_sr_tiles(cuda):
    """BASELINE config 3 shape in small: G on 32x32 LR tiles -> 128x128, fully convolutional D on the SR output
    (VGGStyleDiscriminator128 could not take arbitrary sizes); three ESRGAN iterations stay finite and move D."""
    from collections import OrderedDict as OD
    from image_restoration_amd.models import build_model
    opt = OD(name='t', model_type='ESRGANModel', scale=4, num_gpu=1, manual_seed=0, is_train=True, dist=False, rank=0, world_size=1)
    opt['network_g'] = OD(type='RRDBNet', num_in_ch=3, num_out_ch=3, num_feat=16, num_block=1, num_grow_ch=8)
    opt['network_d'] = OD(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=8, skip_connection=True)
    opt['path'] = OD(pretrain_network_g=None, strict_load_g=True, resume_state=None)
    tr = OD(ema_decay=0.999)
    tr['optim_g'] = OD(type='Adam', lr=1e-4, weight_decay=0, betas=[0.9, 0.99])
    tr['optim_d'] = OD(type='Adam', lr=1e-4, weight_decay=0, betas=[0.9, 0.99])
    tr['scheduler'] = OD(type='MultiStepLR', milestones=[100], gamma=0.5)
    tr['pixel_opt'] = OD(type='L1Loss', loss_weight=1.0, reduction='mean')
    tr['gan_opt'] = OD(type='GANLoss', gan_type='vanilla', real_label_val=1.0, fake_label_val=0.0, loss_weight=0.1)
    tr['net_d_iters'] = 1
    tr['net_d_init_iters'] = 0
    opt['train'] = tr
    model = build_model(opt)
    d0 = model.optimizer_d.flat_p.clone()
    for it in range(1, 4):
        model.update_learning_rate(it)
        model.feed_data({'lq': torch.from_numpy(synth.uniform_input(it, (2, 3, 24, 40))),
                         'gt': torch.from_numpy(synth.uniform_input(100 + it, (2, 3, 96, 160)))})
        model.optimize_parameters(it)
        log = model.get_current_log()
        assert all(np.isfinite(v) for v in log.values()), log
    assert float((model.optimizer_d.flat_p - d0).abs().max()) > 0


def test_batched_spectral_norm_equals_the_per_layer_calls_bit_for_bit(cuda):
    """sr_spectral_norm_fwd_batch_f32 (all spectral-norm layers of a forward, one launch per stage) against sr_spectral_norm_fwd_f32
    per layer: normalised weights, sigma-dependent gradients and the in-place power-iteration buffers, train and eval mode."""
    from image_restoration_amd import hip_autograd as A
    torch.manual_seed(3)
    shapes = [(128, 64, 4, 4), (256, 128, 4, 4), (512, 256, 4, 4), (256, 512, 3, 3), (128, 256, 3, 3), (64, 128, 3, 3), (64, 64, 3, 3), (24, 8, 3, 3)]
    for update in (True, False):
        ws = [torch.randn(*s_, device=cuda).requires_grad_(True) for s_ in shapes]
        ws2 = [w.detach().clone().requires_grad_(True) for w in ws]
        us = [torch.nn.functional.normalize(torch.randn(s_[0], device=cuda), dim=0) for s_ in shapes]
        vs = [torch.nn.functional.normalize(torch.randn(s_[1] * s_[2] * s_[3], device=cuda), dim=0) for s_ in shapes]
        us2, vs2 = [u.clone() for u in us], [v.clone() for v in vs]
        flat = []
        for w, u, v in zip(ws, us, vs):
            flat += [w, u, v]
        outs = A.SpectralNormBatchFn.apply(update, 1e-12, *flat)
        ones = [A.SpectralNormFn.apply(w, u, v, update, 1e-12) for w, u, v in zip(ws2, us2, vs2)]
        gs = [torch.randn_like(o) for o in outs]
        torch.autograd.backward(list(outs), gs)
        torch.autograd.backward(ones, gs)
        for i in range(len(shapes)):
            assert torch.equal(outs[i], ones[i]) and torch.equal(us[i], us2[i]) and torch.equal(vs[i], vs2[i]), (update, i)
            assert torch.equal(ws[i].grad, ws2[i].grad), (update, i)

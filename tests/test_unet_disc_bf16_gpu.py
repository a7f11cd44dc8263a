"""UNetDiscriminatorSN with compute_dtype='bf16' (an extension of an extension: the reference has neither the U-Net
discriminator nor reduced precision).  Op tests compare with float64 computations of the same
bf16-rounded operands (what is tested is the kernel, not the quantisation); the whole network is compared with this
library's fp32 path on the same weights, tolerances stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import image_restoration_amd as ira
from image_restoration_amd import hip_autograd_bf16 as B
from image_restoration_amd import hip_ops as H

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def test_unshuffle_roundtrip_and_order(cuda):
    x = torch.randn(2, 32, 12, 20)
    t = H.nchw_to_cb16(x.to(cuda)).buf
    u = B._unshuffle2(t)
    assert tuple(u.shape) == (2, 8, 6, 10, 16)
    got = H.cb16_to_nchw(H.CB16(u), 128).cpu().double()
    xb = _bf(x)
    for ry in range(2):
        for rx in range(2):
            par = 2 * ry + rx
            assert torch.equal(got[:, par * 32:(par + 1) * 32], xb[:, :, ry::2, rx::2])
    assert torch.equal(B._unshuffle2(u, inverse=True), t)


@pytest.mark.parametrize('cin,cout,n,h,w', [(16, 32, 2, 16, 24), (64, 128, 2, 32, 32), (128, 64, 1, 8, 16)])
def test_conv4x4s2_as_unshuffled_3x3(cuda, cin, cout, n, h, w):
    """Forward, data gradient and weight gradient of the 4x4/s2/p1 conv through ConvFn16 vs float64 autograd of the
    rounded operands: outputs / dx to one bf16 ulp (2^-8 relative + 1e-3 abs), dW to 2e-3 of its max (dz is rounded to bf16
    before the weight gradient, as in every bf16 backward here)."""
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 4, 4, generator=g) * 0.05
    gy = torch.randn(n, cout, h // 2, w // 2, generator=g)
    xr, wr = _bf(x).requires_grad_(True), _bf(wt).requires_grad_(True)
    pre = F.conv2d(xr, wr, None, 2, 1)
    y = F.leaky_relu(pre, 0.2)
    y.backward(_bf(gy))
    xc = H.nchw_to_cb16(x.to(cuda)).buf.requires_grad_(True)
    wc = wt.to(cuda).requires_grad_(True)
    yc = B.ConvFn16.apply(xc, wc, None, 0.2, False)
    got = H.cb16_to_nchw(H.CB16(yc.detach()), cout).cpu().double()
    assert torch.all((got - y.detach()).abs() <= y.detach().abs() * 2 ** -8 + 1e-3)
    yc.backward(H.nchw_to_cb16(gy.to(cuda)).buf)
    dx = H.cb16_to_nchw(H.CB16(xc.grad), cin).cpu().double()
    assert float((dx - xr.grad).abs().max()) <= 2 ** -7 * float(xr.grad.abs().max()) + 1e-3
    assert float((wc.grad.cpu().double() - wr.grad).abs().max()) <= 2e-3 * float(wr.grad.abs().max())


@pytest.mark.parametrize('shape', [(2, 32, 6, 10), (1, 16, 5, 7), (3, 16, 1, 1), (1, 48, 2, 9), (2, 16, 33, 18)])
def test_bilinear2x_bf16(cuda, shape):
    """Forward (2x2 output block per source pixel) and backward (gather over the outputs that touch a source pixel) of the x2
    bilinear resampling against F.interpolate in float64 on the bf16-rounded tensors, odd and degenerate sizes included."""
    n, c, h, w = shape
    x = torch.randn(*shape)
    xr = _bf(x).requires_grad_(True)
    y = F.interpolate(xr, scale_factor=2, mode='bilinear', align_corners=False)
    g = torch.randn(n, c, 2 * h, 2 * w)
    y.backward(_bf(g))
    xc = H.nchw_to_cb16(x.to(cuda)).buf.requires_grad_(True)
    yc = B.Bilinear2xFn16.apply(xc)
    got = H.cb16_to_nchw(H.CB16(yc.detach()), c).cpu().double()
    assert torch.all((got - y.detach()).abs() <= y.detach().abs() * 2 ** -8 + 1e-6)
    yc.backward(H.nchw_to_cb16(g.to(cuda)).buf)
    dx = H.cb16_to_nchw(H.CB16(xc.grad), c).cpu().double()
    assert torch.all((dx - xr.grad).abs() <= xr.grad.abs() * 2 ** -8 + 1e-5)


@pytest.mark.parametrize('c,cout,n,h,w', [(64, 128, 4, 128, 128), (128, 128, 8, 128, 128), (64, 64, 8, 128, 96)])
def test_strided_conv_zero_tap_skipping_is_exact(cuda, c, cout, n, h, w):
    """s2_channels lets the kernel skip the 20 all-zero taps (of 36) of a 4x4/s2 conv embedded in a 3x3 grid: forward and
    data gradient must equal the dense walk over all taps bit for bit (launch sizes that reach the 8-wave tiles)."""
    g = torch.Generator().manual_seed(c + cout)
    u = torch.randn(n, 4 * c // 16, h, w, 16, generator=g).to(torch.bfloat16).to(cuda)      # unshuffled input
    w3 = B._w4_as_w3((torch.randn(cout, c, 4, 4, generator=g) * 0.05).to(cuda))
    src = H.CB16(u)
    pc = H.PackedConvBF16(w3, None)
    dense = H.conv3x3_bf16(src, pc, act_slope=0.2).buf
    assert float(dense.float().abs().max()) > 0
    assert torch.equal(H.conv3x3_bf16(src, pc, act_slope=0.2, s2_channels=c).buf, dense)
    dz = H.CB16(torch.randn(n, cout // 16, h, w, 16, generator=g).to(torch.bfloat16).to(cuda))
    pd = H.PackedConvBF16(w3, None, mode=1)
    assert torch.equal(H.conv3x3_bf16(dz, pd, s2_channels=c, s2_side=1).buf, H.conv3x3_bf16(dz, pd).buf)
    with pytest.raises(Exception):
        H.conv3x3_bf16(src, pc, s2_channels=c // 2)  # not a quarter of the input channels


def test_shared_pass_ops_equal_their_separate_forms(cuda):
    """The fused passes of the bf16 U-Net against the separate ops they replace, bit for bit: skip addition, bilinear
    resampling of a sum, and the fork gradient (sum + way back through the pixel unshuffle + LeakyReLU derivative)."""
    g = torch.Generator().manual_seed(3)
    a = torch.randn(2, 3, 12, 20, 16, generator=g).to(torch.bfloat16).to(cuda)
    b = torch.randn(2, 3, 12, 20, 16, generator=g).to(torch.bfloat16).to(cuda)
    s = B.AddFn16.apply(a, b)
    assert torch.equal(s, (a.float() + b.float()).to(torch.bfloat16))
    ar, br = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = B.Bilinear2xFn16.apply(ar, br)
    assert torch.equal(y, B.Bilinear2xFn16.apply(s))
    gy = torch.randn(2, 3, 24, 40, 16, generator=g).to(torch.bfloat16).to(cuda)
    y.backward(gy)
    sr_ = s.clone().requires_grad_(True)
    B.Bilinear2xFn16.apply(sr_).backward(gy)
    assert torch.equal(ar.grad, sr_.grad) and torch.equal(br.grad, sr_.grad)
    # the resampling gradient carrying the LeakyReLU derivative of the conv that fed it (+ the plain gradient for the skip input):
    # against the float64 resampling gradient of the same bf16 gradient, masked, to one bf16 rounding (the separate passes round twice)
    xm = (torch.randn(2, 3, 12, 20, 16, generator=g) * 0.5).to(torch.bfloat16).to(cuda).requires_grad_(True)
    km = b.clone().requires_grad_(True)
    B.Bilinear2xFn16.apply(xm, km, 0.2).backward(gy)
    assert torch.equal(km.grad, sr_.grad)                       # the skip input: the plain gradient, same bits as the unmasked kernel
    def nchw(t):
        return t.permute(0, 1, 4, 2, 3).reshape(t.size(0), -1, t.size(2), t.size(3))
    ref_in = nchw(s.detach()).double().cpu().requires_grad_(True)
    F.interpolate(ref_in, scale_factor=2, mode='bilinear', align_corners=False).backward(nchw(gy).double().cpu())
    want_m = ref_in.grad * torch.where(nchw(xm.detach()).double().cpu() > 0, 1.0, 0.2)
    got_m = nchw(xm.grad).double().cpu()
    assert torch.all((got_m - want_m).abs() <= want_m.abs() * 2 ** -8 + 1e-6)
    lone = xm.detach().clone().requires_grad_(True)
    B.Bilinear2xFn16.apply(lone, None, 0.2).backward(gy)       # no skip input: masked output only
    assert torch.equal(lone.grad, xm.grad)
    # fork: x [N, C/16, 2h, 2w, 16] -> (x, unshuffle(x)); backward of (g_skip, g_u)
    x = torch.randn(2, 2, 8, 12, 16, generator=g).to(torch.bfloat16).to(cuda).requires_grad_(True)
    xs, u = B.SkipForkFn16.apply(x, 0.2)
    assert torch.equal(xs, x) and torch.equal(u, B._unshuffle2(x.detach()))
    g_skip = torch.randn(2, 2, 8, 12, 16, generator=g).to(torch.bfloat16).to(cuda)
    g_u = torch.randn(2, 8, 4, 6, 16, generator=g).to(torch.bfloat16).to(cuda)
    torch.autograd.backward([xs, u], [g_skip, g_u])
    total = (g_skip.float() + B._unshuffle2(g_u, inverse=True).float()).to(torch.bfloat16)
    want = torch.where(x.detach().float() > 0, total, (total.float() * 0.2).to(torch.bfloat16))
    assert torch.equal(x.grad, want)
    x.grad = None
    xs, u = B.SkipForkFn16.apply(x, 0.2)  # no skip consumer: only the strided conv sends a gradient
    u.backward(g_u)
    back = B._unshuffle2(g_u, inverse=True)
    assert torch.equal(x.grad, torch.where(x.detach().float() > 0, back, (back.float() * 0.2).to(torch.bfloat16)))


def test_conv_chain_with_epilogue_masks_matches_separate_passes(cuda):
    """conv -> LeakyReLU -> conv with the first LeakyReLU's derivative applied in the second conv's data-gradient epilogue
    (input_slope / grad_premasked) against the same chain with separate LeakyReLU-backward passes: identical forward,
    gradients equal up to the one bf16 rounding the fused form saves."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 2, 16, 32, 16, generator=g).to(torch.bfloat16).to(cuda)
    w1 = (torch.randn(32, 32, 3, 3, generator=g) * 0.08).to(cuda)
    w2 = (torch.randn(16, 32, 3, 3, generator=g) * 0.08).to(cuda)
    gy = torch.randn(2, 1, 16, 32, 16, generator=g).to(torch.bfloat16).to(cuda)
    res = []
    for fused in (False, True):
        xr, a, b = x.clone().requires_grad_(True), w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
        h1 = B.ConvFn16.apply(xr, a, None, 0.2, False, False, 1.0, fused)
        y = B.ConvFn16.apply(h1, b, None, 0.2, False, False, 0.2 if fused else 1.0, False)
        y.backward(gy)
        res.append((y.detach(), xr.grad, a.grad, b.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][3], res[1][3])
    for u, v in zip(res[0][1:3], res[1][1:3]):
        d = (u.float() - v.float()).abs()
        assert float(d.max()) <= 2 ** -6 * float(u.float().abs().max()) and float(d.norm() / u.float().norm()) < 4e-3


@pytest.mark.parametrize('nf,h,w,skip', [(16, 32, 48, True), (64, 64, 64, True), (32, 40, 24, False)])
def test_unet_bf16_vs_fp32_path(cuda, nf, h, w, skip):
    """Logits and parameter gradients of the bf16 network against the fp32 HIP network (itself checked against the oracle in
    tests/test_unet_disc_gpu.py) on the same weights: logits within 3 % of their range, every gradient within relative-L2
    8e-2 and cosine >= 0.995 (10 layers of bf16 activations and activation gradients)."""
    torch.manual_seed(7)
    d32 = ira.build_network(dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=nf, skip_connection=skip)).to(cuda).train()
    d16 = ira.build_network(dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=nf, skip_connection=skip,
                                 compute_dtype='bf16')).to(cuda).train()
    d16.load_state_dict(d32.state_dict())
    x = torch.rand(2, 3, h, w, device=cuda)
    tgt = torch.rand(2, 1, h, w, device=cuda)
    outs, grads = {}, {}
    for name, net in (('fp32', d32), ('bf16', d16)):
        xi = x.clone().requires_grad_(True)
        y = net(xi)
        F.binary_cross_entropy_with_logits(y, tgt).backward()  # loss on the fp32 logits: plain torch, not the path under test
        outs[name] = y.detach()
        grads[name] = [xi.grad] + [p.grad for p in net.parameters()]
    assert outs['bf16'].dtype == torch.float32 and outs['bf16'].shape == outs['fp32'].shape
    rng = float(outs['fp32'].max() - outs['fp32'].min())
    assert float((outs['bf16'] - outs['fp32']).abs().max()) <= 3e-2 * rng
    names = ['x'] + [k for k, _ in d32.named_parameters()]
    for name, a, b in zip(names, grads['fp32'], grads['bf16']):
        assert torch.isfinite(b).all(), name
        rel = float((a - b).norm() / (a.norm() + 1e-20))
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-20))
        # without skip connections every gradient funnels through the 512-channel bottleneck (5 x 3 pixels here) and bf16 storage noise
        # is 2-3x larger (measured 0.17 / 0.985 on the input gradient) — the same network agrees with the float64 model of bf16
        # storage to 1.2e-2 (next test), so the wider bound is about bf16, not about the kernels
        assert (rel <= 8e-2 and cos >= 0.995) if skip else (rel <= 0.3 and cos >= 0.96), (name, rel, cos)
    # power-iteration buffers advanced identically (spectral norm stays fp32)
    assert torch.allclose(d16.conv3.weight_u, d32.conv3.weight_u, atol=1e-6)


def test_bn_lrelu_bf16(cuda):
    """BatchNorm2d(train) + LeakyReLU on CB16 vs torch on the bf16-rounded input: output to one bf16 ulp, running statistics
    and parameter gradients 1e-3 relative, dx within 2^-7 of its max."""
    torch.manual_seed(0)
    n, c, h, w = 4, 48, 10, 12
    x = torch.randn(n, c, h, w) * 2 + 0.5
    bn = torch.nn.BatchNorm2d(c).double().train()
    bn.weight.data.uniform_(0.5, 1.5)
    bn.bias.data.uniform_(-0.5, 0.5)
    xr = _bf(x).requires_grad_(True)
    y = F.leaky_relu(bn(xr), 0.2)
    g = torch.randn(n, c, h, w)
    y.backward(_bf(g))
    gamma = bn.weight.detach().float().to(cuda).requires_grad_(True)
    beta = bn.bias.detach().float().to(cuda).requires_grad_(True)
    rm, rv = torch.zeros(c, device=cuda), torch.ones(c, device=cuda)
    xc = H.nchw_to_cb16(x.to(cuda)).buf.requires_grad_(True)
    yc = B.BNLReLUFn16.apply(xc, gamma, beta, rm, rv, True, 0.1, 1e-5, 0.2)
    got = H.cb16_to_nchw(H.CB16(yc.detach()), c).cpu().double()
    assert torch.all((got - y.detach()).abs() <= y.detach().abs() * 2 ** -8 + 1e-3)
    assert torch.allclose(rm.cpu().double(), bn.running_mean, rtol=1e-3, atol=1e-4)
    assert torch.allclose(rv.cpu().double(), bn.running_var, rtol=1e-3, atol=1e-4)
    yc.backward(H.nchw_to_cb16(g.to(cuda)).buf)
    dx = H.cb16_to_nchw(H.CB16(xc.grad), c).cpu().double()
    assert float((dx - xr.grad).abs().max()) <= 2 ** -7 * float(xr.grad.abs().max())
    assert float((gamma.grad.cpu().double() - bn.weight.grad).abs().max()) <= 2e-3 * float(bn.weight.grad.abs().max())
    assert float((beta.grad.cpu().double() - bn.bias.grad).abs().max()) <= 2e-3 * float(bn.bias.grad.abs().max())


def test_vgg128_bf16_vs_fp32_path(cuda):
    """VGGStyleDiscriminator128 bf16 vs the fp32 HIP network (pinned to the reference by golden g_g_vgg128) on the same
    weights, train mode.  The bound is the noise bf16 storage itself causes in this network: a float64 copy of the
    architecture with nothing but a bf16 rounding of activations, weights and activation gradients between ops is already
    0.27 relative-L2 / cosine 0.966 off its exact gradient at conv0_0 and 0.15 / 0.99 at conv4_0 (19 conv / BatchNorm
    stages, each re-normalising by batch statistics; measured with a torch-CPU simulation when this test was written), and
    the HIP path measures 0.24 / 0.97 and 0.17 / 0.985 there.  Asserted: logits within 5 % of their spread; every gradient
    relative-L2 <= 0.35 and cosine >= 0.95; the last four stages <= 0.12 / >= 0.99; running statistics within 2e-2."""
    torch.manual_seed(4)
    d32 = ira.build_network(dict(type='VGGStyleDiscriminator128', num_in_ch=3, num_feat=16)).to(cuda).train()
    d16 = ira.build_network(dict(type='VGGStyleDiscriminator128', num_in_ch=3, num_feat=16, compute_dtype='bf16')).to(cuda).train()
    d16.load_state_dict(d32.state_dict())
    x = torch.rand(6, 3, 128, 128, device=cuda)
    outs, grads = {}, {}
    for name, net in (('fp32', d32), ('bf16', d16)):
        y = net(x)
        F.binary_cross_entropy_with_logits(y, torch.ones_like(y) * 0.3).backward()
        outs[name] = y.detach()
        grads[name] = [p.grad for p in net.parameters()]
    spread = float(outs['fp32'].std()) + float(outs['fp32'].abs().mean())
    assert float((outs['bf16'] - outs['fp32']).abs().max()) <= 5e-2 * spread
    for (k, _), a, b in zip(d32.named_parameters(), grads['fp32'], grads['bf16']):
        assert torch.isfinite(b).all(), k
        rel = float((a - b).norm() / (a.norm() + 1e-20))
        cos = float((a * b).sum() / (a.norm() * b.norm() + 1e-20))
        assert rel <= 0.35 and cos >= 0.95, (k, rel, cos)
        if k.split('.')[0] in ('bn4_1', 'linear1', 'linear2'):
            assert rel <= 0.12 and cos >= 0.99, (k, rel, cos)
    assert torch.allclose(d16.bn4_1.running_var, d32.bn4_1.running_var, rtol=2e-2, atol=1e-4)
    assert int(d16.bn0_1.num_batches_tracked) == 1


def test_c3_training_config_runs_all_bf16(cuda):
    """training_config/train_rrdbnet_esrgan_x4_mi355x_bf16_unet.yml (BASELINE configs[2]: bf16 generator + bf16
    UNetDiscriminatorSN) with the networks shrunk: three optimize_parameters iterations, finite losses of the right keys,
    parameters actually move, EMA follows."""
    import os
    from image_restoration_amd.models import build_model
    from image_restoration_amd.utils.options import parse
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    opt = parse(os.path.join(root, 'training_config', 'train_rrdbnet_esrgan_x4_mi355x_bf16_unet.yml'), root, is_train=True)
    opt['dist'], opt['rank'], opt['world_size'], opt['num_gpu'] = False, 0, 1, 1
    assert opt['network_g']['compute_dtype'] == 'bf16' and opt['network_d']['compute_dtype'] == 'bf16'
    opt['network_g'].update(num_feat=32, num_block=1, num_grow_ch=16)
    opt['network_d']['num_feat'] = 16
    model = build_model(opt)
    assert model.net_g.compute_dtype == 'bf16' and model.net_d.compute_dtype == 'bf16'
    w0 = model.net_g.conv_last.weight.detach().clone()
    d0 = model.net_d.conv9.weight.detach().clone()
    g = torch.Generator().manual_seed(0)
    for it in range(1, 4):
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_data({'lq': torch.rand(2, 3, 16, 24, generator=g), 'gt': torch.rand(2, 3, 64, 96, generator=g)})
        model.optimize_parameters(it)
        log = model.get_current_log()
        assert set(log) == {'l_g_pix', 'l_g_gan', 'l_d_real', 'l_d_fake', 'out_d_real', 'out_d_fake'}
        assert all(np.isfinite(v) for v in log.values()), log
    assert not torch.equal(model.net_g.conv_last.weight, w0) and not torch.equal(model.net_d.conv9.weight, d0)
    assert not torch.equal(model.net_g_ema.conv_last.weight, w0)


@pytest.mark.parametrize('skip', [True, False])
def test_unet_bf16_equals_a_float64_model_of_bf16_storage(cuda, skip):
    """The bf16 U-Net (fused fork / add-bilinear / epilogue-mask passes) against oracle/bf16_sim.py — float64 arithmetic with a
    bf16 round trip wherever the HIP path stores a tensor — on the effective (spectrally normalised) weights, eval mode:
    logits to 5e-3 relative L2, input gradient and the gradients of the plain parameters (conv0, conv9) to 3e-2; against the
    fp32 path the same quantities differ by 2-8 % (test above)."""
    from oracle.bf16_sim import unet_forward_bf16_storage
    torch.manual_seed(11)
    net = ira.build_network(dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=16, skip_connection=skip, compute_dtype='bf16')).to(cuda)
    with torch.no_grad():
        for _ in range(12):  # let the power iterations settle so that the normalised weights (and logits) are O(1)
            net(torch.rand(1, 3, 16, 16, device=cuda))
    net.eval()
    w = {}
    for i in range(10):
        conv = getattr(net, f'conv{i}')
        wt = conv.weight if i in (0, 9) else conv.weight()  # SN layers: the normalised weight the forward uses (eval: no update)
        w[f'conv{i}.weight'] = wt.detach().cpu().double().requires_grad_(True)
        if i in (0, 9):
            w[f'conv{i}.bias'] = conv.bias.detach().cpu().double().requires_grad_(True)
    x = torch.rand(2, 3, 48, 64)
    xr = x.double().requires_grad_(True)
    yr = unet_forward_bf16_storage(xr, w, skip_connection=skip)
    R = torch.randn(yr.shape, generator=torch.Generator().manual_seed(1))
    (yr * R.double()).sum().backward()
    xc = x.to(cuda).requires_grad_(True)
    y = net(xc)
    (y * R.to(cuda)).sum().backward()

    def rel(a, b):
        return float((a.double().cpu() - b).norm() / b.norm())
    assert rel(y.detach(), yr.detach()) < 5e-3
    assert rel(xc.grad, xr.grad) < 3e-2
    for name, p in (('conv0.weight', net.conv0.weight), ('conv0.bias', net.conv0.bias), ('conv9.weight', net.conv9.weight),
                    ('conv9.bias', net.conv9.bias)):
        assert rel(p.grad, w[name].grad) < 3e-2, name


@pytest.mark.parametrize('n,cin,cout,h,w', [
    (2, 64, 64, 32, 48),      # per-tile kernel, small launch (4-wave tiles)
    (4, 128, 128, 64, 96),    # per-tile kernel, 8-wave tiles, two cout groups
    (12, 16, 64, 512, 512),   # streaming kernel (few input channels on a large grid): the U-Net's conv0 at its training size
    (3, 48, 32, 18, 22),      # ragged: partly filled tiles, 32 couts
])
def test_conv_stores_its_output_pixel_unshuffled_only(cuda, n, cin, cout, h, w):
    """sr_conv3x3_desc.out_unshuffle2: the epilogue writes out[n][(2 ry + rx) CB + cb][y / 2][x / 2] — bit for bit the pixel unshuffle
    (sr_cb16_unshuffle2_bf16) of the plainly stored result, for the per-tile and the streaming kernel."""
    g = torch.Generator().manual_seed(n + cin)
    x = H.CB16(torch.randn(n, cin // 16, h, w, 16, generator=g).to(torch.bfloat16).to(cuda))
    pc = H.PackedConvBF16((torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(cuda), (torch.randn(cout, generator=g) * 0.1).to(cuda))
    plain = H.conv3x3_bf16(x, pc, act_slope=0.2).buf
    u = H.conv3x3_bf16(x, pc, act_slope=0.2, out_unshuffle2=True).buf
    assert u.shape == (n, 4 * cout // 16, h // 2, w // 2, 16)
    assert torch.equal(u, B._unshuffle2(plain))


def test_readers_of_unshuffled_only_activations_equal_their_plain_forms(cuda):
    """sr_cb16_add_u2_bf16, sr_bilinear2x_fwd_u2_bf16 and sr_cb16_fork_bwd_u2_bf16 read an activation that only exists
    pixel-unshuffled: same bits as the plain-layout kernels on the shuffled-back tensor; the autograd functions pass gradients of
    such tensors in the plain layout under the unshuffled shape."""
    g = torch.Generator().manual_seed(11)
    n, cb, h2, w2 = 2, 3, 12, 20
    a = torch.randn(n, cb, h2, w2, 16, generator=g).to(torch.bfloat16).to(cuda)
    b = torch.randn(n, cb, h2, w2, 16, generator=g).to(torch.bfloat16).to(cuda)
    bu = B._unshuffle2(b)
    assert torch.equal(B.AddFn16.apply(a, bu, True), B.AddFn16.apply(a, b))
    assert torch.equal(B.Bilinear2xFn16.apply(a, bu, 1.0, True), B.Bilinear2xFn16.apply(a, b))
    # gradients: the skip input's gradient comes back plain-layout under bu's shape
    ar, bur = a.clone().requires_grad_(True), bu.clone().requires_grad_(True)
    gy = torch.randn(n, cb, 2 * h2, 2 * w2, 16, generator=g).to(torch.bfloat16).to(cuda)
    B.Bilinear2xFn16.apply(ar, bur, 0.2, True).backward(gy)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    B.Bilinear2xFn16.apply(a2, b2, 0.2).backward(gy)
    assert torch.equal(ar.grad, a2.grad) and bur.grad.shape == bu.shape and torch.equal(bur.grad.view(b.shape), b2.grad)
    # fork: u -> (skip handle, strided-conv input); backward of (g_skip plain-under-u-shape, g_u)
    x = torch.randn(n, cb, h2, w2, 16, generator=g).to(torch.bfloat16).to(cuda)
    u = B._unshuffle2(x).requires_grad_(True)
    s_, c_ = B.ForkU2Fn16.apply(u, 0.2)
    assert s_.data_ptr() == u.data_ptr() and c_.data_ptr() == u.data_ptr()      # no copy
    g_skip = torch.randn(n, cb, h2, w2, 16, generator=g).to(torch.bfloat16).to(cuda)
    g_u = torch.randn(n, 4 * cb, h2 // 2, w2 // 2, 16, generator=g).to(torch.bfloat16).to(cuda)
    torch.autograd.backward([s_, c_], [g_skip.view(u.shape), g_u])
    xr = x.clone().requires_grad_(True)
    xs, xu = B.SkipForkFn16.apply(xr, 0.2)
    torch.autograd.backward([xs, xu], [g_skip, g_u])
    assert torch.equal(u.grad.view(x.shape), xr.grad)


@pytest.mark.parametrize('n,cin,cout,h,w', [(2, 64, 64, 32, 48), (4, 128, 64, 64, 96), (1, 32, 32, 10, 14)])
def test_skip_added_in_the_conv_epilogue_keeps_the_activation_sign_recoverable(cuda, n, cin, cout, h, w):
    """sr_conv3x3_desc.res1_u2 + res1_keep_sign: out = bf16(LeakyReLU(conv) + x0) with x0 read from its pixel-unshuffled form and the
    rounding nudged by at most one bf16 ulp so that sign(out - x0) == sign(conv) for EVERY element — checked against the separately
    computed activation — which is what lets sr_lrelu_bwd_diff_u2_bf16 replace the stored activation in the backward (bit-identical
    to sr_lrelu_bwd_bf16 on the real activation).  The skip is made much larger than the activation so that absorbed sums are common."""
    import ctypes as C
    from image_restoration_amd import _lib
    g = torch.Generator().manual_seed(cin + h)
    x = H.CB16(torch.randn(n, cin // 16, h, w, 16, generator=g).to(torch.bfloat16).to(cuda))
    pc = H.PackedConvBF16((torch.randn(cout, cin, 3, 3, generator=g) * 0.01).to(cuda), None)
    x0 = (torch.randn(n, cout // 16, h, w, 16, generator=g) * 40).to(torch.bfloat16).to(cuda)
    x0[0, 0, :2] = 0.0                                        # zeros (both signs) among the skip values
    x0[0, 0, 1] = -0.0
    x0u = B._unshuffle2(x0)
    act = H.conv3x3_bf16(x, pc, act_slope=0.2).buf            # the activation, stored on its own (what the fused form never stores)
    out = H.conv3x3_bf16(x, pc, act_slope=0.2, res1=H.CB16(x0u), beta1=1.0, res1_u2=True, res1_keep_sign=True).buf
    plain = H.conv3x3_bf16(x, pc, act_slope=0.2, res1=H.CB16(x0), beta1=1.0).buf          # ordinary rounding of the same sum
    diff = out.float() - x0.float()
    # (the bf16-stored activation has the sign of the fp32 one except where it rounded to zero: compare where it is non-zero)
    nz = act.float() != 0
    assert torch.equal((diff > 0)[nz], (act.float() > 0)[nz])
    assert not bool(((diff > 0) & (act.float() < 0)).any())
    changed = out != plain
    assert 0 < int(changed.sum()) < 0.5 * out.numel()         # the nudge is exercised, and it is the exception
    one_ulp = (out.view(torch.int16).int() - plain.view(torch.int16).int()).abs() <= 1
    assert bool(one_ulp[changed].all()) and bool((plain == x0)[changed].all())
    # backward mask from (out, x0_u2) == mask from the activation
    gy = torch.randn(n, cout // 16, h, w, 16, generator=g).to(torch.bfloat16).to(cuda)
    lib = _lib.load()
    dz = torch.empty_like(gy)
    _lib.check(lib.sr_lrelu_bwd_diff_u2_bf16(gy.data_ptr(), out.data_ptr(), x0u.data_ptr(), dz.data_ptr(), 0.2, n, cout // 16, h // 2, w // 2,
                                             None), 'sr_lrelu_bwd_diff_u2_bf16')
    want = torch.where(diff > 0, gy, (gy.float() * 0.2).to(torch.bfloat16))
    assert torch.equal(dz, want)
    # autograd wiring: ConvFn16(skip_u2=...) against conv -> AddFn16 with the same weights
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(cuda).requires_grad_(True)
    xin = x.buf.clone().requires_grad_(True)
    sk = x0u.clone().requires_grad_(True)
    y1 = B.ConvFn16.apply(xin, wt, None, 0.2, False, False, 1.0, False, False, sk)
    y1.backward(gy)
    wt2, xin2, sk2 = wt.detach().clone().requires_grad_(True), x.buf.clone().requires_grad_(True), x0u.clone().requires_grad_(True)
    y2 = B.AddFn16.apply(B.ConvFn16.apply(xin2, wt2, None, 0.2, False), sk2, True)
    y2.backward(gy)
    assert float((y1.float() - y2.float()).abs().max()) <= float(y2.float().abs().max()) * 2 ** -7
    assert torch.equal(sk.grad, sk2.grad)
    for a_, b_ in ((wt.grad, wt2.grad), (xin.grad.float(), xin2.grad.float())):
        assert float((a_ - b_).norm()) <= 2e-2 * float(b_.norm()) + 1e-6   # masks differ only where the activation is within rounding of 0

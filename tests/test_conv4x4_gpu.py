"""4x4 / stride 2 / pad 1 convolution of the discriminators (discriminator_arch.py:22-43) on the HIP path:
forward, data gradient and weight gradient against PyTorch-CPU (F.conv2d and its autograd), 1e-4 relative."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from image_restoration_amd import hip_ops as H
from image_restoration_amd.utils import synth

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize('cin,cout,h,w,bias', [(8, 8, 16, 16, False), (64, 64, 32, 48, False), (64, 128, 20, 36, True),
                                                (128, 256, 16, 70, False), (16, 32, 10, 6, True), (3, 20, 12, 14, True),
                                                # small maps -> narrow tiles over the stacked batch (conv_f32.hip: launch_small)
                                                (64, 64, 32, 32, True), (64, 128, 16, 16, False), (128, 64, 8, 8, True),
                                                (64, 64, 20, 24, False), (64, 64, 4, 4, True),
                                                # wide: 16 x 16 tile groups per parity pass in one launch
                                                (512, 512, 8, 8, False), (160, 288, 8, 12, True)])
def test_conv4x4s2_fwd_dgrad_wgrad(cuda, cin, cout, h, w, bias):
    n = 5 if w <= 32 else 2
    rng = np.random.default_rng(cin * 7 + cout)
    wt = torch.from_numpy((rng.standard_normal((cout, cin, 4, 4)) * 0.05).astype(np.float32)).requires_grad_(True)
    bs = torch.from_numpy((rng.standard_normal((cout,)) * 0.1).astype(np.float32)).requires_grad_(True) if bias else None
    x = torch.from_numpy(synth.signed_input(cin + h, (n, cin, h, w))).requires_grad_(True)
    pre = F.conv2d(x, wt, bs, stride=2, padding=1)
    y = F.leaky_relu(pre, 0.2)
    gy = torch.from_numpy(synth.signed_input(cout + w, tuple(y.shape)))
    gpre, = torch.autograd.grad(y, pre, gy, retain_graph=True)
    pre.backward(gpre)

    src = H.nchw_to_cb8(x.detach().to(cuda))
    pc = H.PackedConv4x4s2(wt.detach().to(cuda), bs.detach().to(cuda) if bias else None)
    out = H.conv4x4s2(src, pc, act_slope=0.2)
    assert (out.h, out.w) == tuple(y.shape[2:])
    assert _rel(H.cb8_to_nchw(out, cout), y) < 1e-4

    gp = H.nchw_to_cb8(gpre.to(cuda))
    pcd = H.PackedConv4x4s2(wt.detach().to(cuda), None, mode=1)
    dx = H.conv4x4s2_dgrad(gp, pcd, h, w)
    assert _rel(H.cb8_to_nchw(dx, cin), x.grad) < 1e-4

    dw, db = H.conv4x4s2_wgrad(src, gp, cout, cin, want_bias=bias)
    assert _rel(dw, wt.grad) < 1e-4
    if bias:
        assert _rel(db, bs.grad) < 1e-4


@pytest.mark.parametrize('cin,cout,n,h,w', [(64, 64, 5, 16, 16), (128, 128, 7, 8, 8), (64, 64, 9, 4, 4), (64, 128, 3, 12, 10),
                                            (64, 64, 1, 16, 16), (72, 64, 4, 6, 16)])
def test_conv3x3_small_maps_stacked_tiles(cuda, cin, cout, n, h, w):
    """3x3 convs of the discriminator's 16x16 ... 4x4 layers run on narrow tiles over the vertically stacked batch: forward
    with residual, and the data gradient with the LeakyReLU mask, against PyTorch-CPU (1e-4 relative)."""
    rng = np.random.default_rng(cin + cout + n)
    wt = torch.from_numpy((rng.standard_normal((cout, cin, 3, 3)) * 0.05).astype(np.float32)).requires_grad_(True)
    bs = torch.from_numpy((rng.standard_normal((cout,)) * 0.1).astype(np.float32))
    x = torch.from_numpy(synth.signed_input(cin + h, (n, cin, h, w))).requires_grad_(True)
    res = torch.from_numpy(synth.signed_input(7, (n, cout, h, w)))
    pre = F.conv2d(x, wt, bs, padding=1)
    y = 0.5 * F.leaky_relu(pre, 0.2) + 0.25 * res
    src = H.nchw_to_cb8(x.detach().to(cuda))
    pc = H.PackedConv(wt.detach().to(cuda), bs.to(cuda))
    out = H.conv3x3(src, pc, act_slope=0.2, alpha=0.5, res1=H.nchw_to_cb8(res.to(cuda)), beta1=0.25)
    assert _rel(H.cb8_to_nchw(out, cout), y) < 1e-4
    # data gradient of the pre-activation, masked by an activation map (as the backward of a preceding LeakyReLU)
    gpre = torch.from_numpy(synth.signed_input(cout + w, (n, cout, h, w)))
    act = torch.from_numpy(synth.signed_input(11, (n, cin, h, w)))
    pre.backward(gpre)
    want = torch.where(act > 0, x.grad, 0.2 * x.grad)
    pcd = H.PackedConv(wt.detach().to(cuda), None, mode=1)
    dx = H.conv3x3(H.nchw_to_cb8(gpre.to(cuda)), pcd, mask=H.nchw_to_cb8(act.to(cuda)))
    cin_pad = (cin + 7) // 8 * 8
    assert _rel(H.cb8_to_nchw(dx, cin), want) < 1e-4 and dx.channels == cin_pad

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
                                                (128, 256, 16, 70, False), (16, 32, 10, 6, True), (3, 20, 12, 14, True)])
def test_conv4x4s2_fwd_dgrad_wgrad(cuda, cin, cout, h, w, bias):
    n = 2
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

"""BatchNorm2d (+ LeakyReLU) of small activations in ONE launch (csrc/bn_small.h) against the general two-stage path it short-cuts
(n * h * w <= 32768 pixels: the 32 x 32 ... 4 x 4 levels of VGGStyleDiscriminator128, discriminator_arch.py:23-49, on the reference
recipe's batch of 32) and against torch.nn.BatchNorm2d in float64.  Both HIP paths sum in fixed orders (bit-reproducible); they are
not bit-identical to each other, so the comparison is at fp32 / bf16 rounding level.  Channel counts that do not fill a channel block
(padding channels must come out as zeros), the pixel-count threshold and both dtypes are covered."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from image_restoration_amd import _lib
from image_restoration_amd import hip_autograd as A
from image_restoration_amd import hip_autograd_bf16 as B
from image_restoration_amd import hip_ops as H

pytestmark = pytest.mark.gpu


@pytest.fixture
def switch():
    lib = _lib.load()
    lib.sr_dev_set_bn_small.argtypes = [C.c_int]
    lib.sr_dev_set_bn_small.restype = None
    yield lib.sr_dev_set_bn_small
    lib.sr_dev_set_bn_small(1)


def _run(dtype, x, g, gamma0, beta0, dev):
    gamma, beta = gamma0.clone().to(dev).requires_grad_(True), beta0.clone().to(dev).requires_grad_(True)
    c = gamma.numel()
    rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
    if dtype == 'bf16':
        xc = H.nchw_to_cb16(x.to(dev)).buf.requires_grad_(True)
        yc = B.BNLReLUFn16.apply(xc, gamma, beta, rm, rv, True, 0.1, 1e-5, 0.2)
        yc.backward(H.nchw_to_cb16(g.to(dev)).buf)
        return dict(y=yc.detach(), dx=xc.grad, dgamma=gamma.grad, dbeta=beta.grad, rm=rm, rv=rv,
                    y_nchw=H.cb16_to_nchw(H.CB16(yc.detach()), c), dx_nchw=H.cb16_to_nchw(H.CB16(xc.grad), c))
    xc = H.nchw_to_cb8(x.to(dev)).buf.requires_grad_(True)
    yc = A.BNLReLUFn.apply(xc, gamma, beta, rm, rv, True, 0.1, 1e-5, 0.2)
    yc.backward(H.nchw_to_cb8(g.to(dev)).buf)
    return dict(y=yc.detach(), dx=xc.grad, dgamma=gamma.grad, dbeta=beta.grad, rm=rm, rv=rv,
                y_nchw=H.cb8_to_nchw(H.CB8(yc.detach()), c), dx_nchw=H.cb8_to_nchw(H.CB8(xc.grad), c))


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
@pytest.mark.parametrize('n,c,h,w', [(32, 512, 4, 4), (32, 256, 16, 16), (5, 20, 7, 9), (3, 4, 5, 5), (32, 64, 32, 32), (2, 16, 128, 128)])
def test_one_launch_batchnorm_equals_the_general_path_and_torch(cuda, switch, dtype, n, c, h, w):
    torch.manual_seed(n + c)
    x = torch.randn(n, c, h, w) * 1.5 + 0.3
    g = torch.randn(n, c, h, w)
    gamma0, beta0 = torch.rand(c) + 0.5, torch.rand(c) - 0.5
    switch(1)
    one = _run(dtype, x, g, gamma0, beta0, cuda)
    switch(0)
    gen = _run(dtype, x, g, gamma0, beta0, cuda)
    tol = 2e-5 if dtype == 'fp32' else 1.0 / 128      # fp32 summation order / one bf16 ulp of the stored tensors
    for k in ('y', 'dx'):
        a, b = one[k].float(), gen[k].float()
        assert float((a - b).abs().max()) <= tol * float(b.abs().max()) + 1e-6, k
    for k in ('dgamma', 'dbeta', 'rm', 'rv'):
        assert float((one[k] - gen[k]).abs().max()) <= 2e-5 * float(gen[k].abs().max()) + 1e-6, k
    # padding channels of the last block are zeros in both outputs
    cbw = 8 if dtype == 'fp32' else 16
    if c % cbw:
        flat = one['y'].float().permute(0, 1, 4, 2, 3).reshape(n, -1, h, w)
        assert float(flat[:, c:].abs().max()) == 0.0 and float(one['dx'].float().permute(0, 1, 4, 2, 3).reshape(n, -1, h, w)[:, c:].abs().max()) == 0.0
    # torch.nn.BatchNorm2d in float64 on the (rounded) input
    xr = (x.to(torch.bfloat16).double() if dtype == 'bf16' else x.double()).requires_grad_(True)
    bn = torch.nn.BatchNorm2d(c).double().train()
    bn.weight.data.copy_(gamma0.double())
    bn.bias.data.copy_(beta0.double())
    y = F.leaky_relu(bn(xr), 0.2)
    y.backward(g.to(torch.bfloat16).double() if dtype == 'bf16' else g.double())
    rel = 1e-4 if dtype == 'fp32' else 2 ** -7
    assert float((one['y_nchw'].cpu().double() - y.detach()).abs().max()) <= rel * float(y.detach().abs().max()) + 1e-5
    assert float((one['dx_nchw'].cpu().double() - xr.grad).abs().max()) <= rel * float(xr.grad.abs().max()) + 1e-6
    assert torch.allclose(one['rm'].cpu().double(), bn.running_mean, rtol=1e-3, atol=1e-5)
    assert torch.allclose(one['rv'].cpu().double(), bn.running_var, rtol=1e-3, atol=1e-5)
    assert float((one['dgamma'].cpu().double() - bn.weight.grad).abs().max()) <= 3e-3 * float(bn.weight.grad.abs().max()) + 1e-5
    assert float((one['dbeta'].cpu().double() - bn.bias.grad).abs().max()) <= 3e-3 * float(bn.bias.grad.abs().max()) + 1e-5


def test_one_launch_batchnorm_is_bit_reproducible(cuda):
    torch.manual_seed(0)
    x, g = torch.randn(32, 128, 8, 8), torch.randn(32, 128, 8, 8)
    gamma0, beta0 = torch.rand(128) + 0.5, torch.rand(128) - 0.5
    a = _run('bf16', x, g, gamma0, beta0, cuda)
    b = _run('bf16', x, g, gamma0, beta0, cuda)
    for k in ('y', 'dx', 'dgamma', 'dbeta', 'rm', 'rv'):
        assert torch.equal(a[k], b[k]), k

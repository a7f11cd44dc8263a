"""Whole-network drivers of the bf16 U-Net discriminator (sr_unet_forward_bf16 / sr_unet_backward_bf16, include/sr_hip.h) against the
per-layer route of rounds 2-3 (hip_autograd_bf16.py, checked against the oracle and the float64 bf16-storage model in
tests/test_unet_disc_bf16_gpu.py): the same launches in the same order, so logits, dL/dx, every parameter gradient (through the
spectral normalisation's backward) and the power-iteration buffers must be BIT-identical."""
import copy

import pytest
import torch

import image_restoration_amd as ira
from image_restoration_amd.utils import synth

pytestmark = pytest.mark.gpu


def _net(dev, nf, skip, seed=3):
    torch.manual_seed(seed)
    net = ira.build_network(dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=nf, skip_connection=skip, compute_dtype='bf16')).to(dev)
    return net.train()


def _run(net, x, wgt, layers, need_x=True):
    net.zero_grad(set_to_none=True)
    xx = x.clone().requires_grad_(need_x)
    out = net.forward_layers(xx) if layers else net(xx)
    (out * wgt).sum().backward()
    return out.detach().clone(), (xx.grad.clone() if need_x else None), {k: (p.grad.clone() if p.grad is not None else None)
                                                                          for k, p in net.named_parameters()}


@pytest.mark.parametrize('nf,skip,shape', [(16, True, (2, 3, 64, 96)), (16, False, (1, 3, 40, 24)), (64, True, (2, 3, 128, 128)),
                                           (32, True, (3, 3, 8, 16))])
def test_unet_driver_is_bit_identical_to_the_per_layer_route(cuda, nf, skip, shape):
    a = _net(cuda, nf, skip)
    b = copy.deepcopy(a)
    assert a.use_driver
    x = torch.from_numpy(synth.uniform_input(5, shape)).to(cuda)
    wgt = torch.from_numpy(synth.signed_input(6, (shape[0], 1, shape[2], shape[3]))).to(cuda)
    for it in range(2):   # the second pass starts from moved power-iteration vectors
        oa, gxa, gpa = _run(a, x, wgt, False)
        ob, gxb, gpb = _run(b, x, wgt, True)
        assert torch.equal(oa, ob), it
        assert torch.equal(gxa, gxb), it
        for k in gpa:
            assert torch.equal(gpa[k], gpb[k]), (it, k)
        for (k, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
            assert torch.equal(ba, bb), (it, k)
    # input that needs no gradient (the critic phase: fake.detach())
    oa, _, gpa = _run(a, x, wgt, False, need_x=False)
    ob, _, gpb = _run(b, x, wgt, True, need_x=False)
    assert torch.equal(oa, ob) and all(torch.equal(gpa[k], gpb[k]) for k in gpa)
    # a frozen discriminator (generator phase): input gradient only
    for net in (a, b):
        for p in net.parameters():
            p.requires_grad = False
    oa, gxa, gpa = _run(a, x, wgt, False)
    ob, gxb, gpb = _run(b, x, wgt, True)
    assert torch.equal(oa, ob) and torch.equal(gxa, gxb) and all(v is None for v in gpa.values())
    # eval mode: no power iteration, buffers stay
    a.eval(), b.eval()
    before = {k: v.clone() for k, v in a.named_buffers()}
    with torch.no_grad():
        assert torch.equal(a(x), b.forward_layers(x))
    assert all(torch.equal(v, before[k]) for k, v in a.named_buffers())


def test_unet_driver_rejects_what_it_cannot_run(cuda):
    net = _net(cuda, 16, True)
    with pytest.raises(AssertionError):
        net(torch.zeros(1, 3, 12, 16, device=cuda))      # not a multiple of 8
    with pytest.raises(ira._lib.SrHipError):
        net(torch.zeros(1, 3, 16, 16))                    # CPU tensor: no fallback
    # a partially frozen network takes the per-layer route and still trains the rest
    net.conv4.weight_orig.requires_grad = False
    x = torch.rand(1, 3, 16, 16, device=cuda)
    net(x).sum().backward()
    assert net.conv4.weight_orig.grad is None and net.conv5.weight_orig.grad is not None

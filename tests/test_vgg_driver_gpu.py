"""Whole-network drivers of the VGG-style discriminators (sr_vgg_forward_* / sr_vgg_backward_* / sr_vgg_apply_stats_*,
include/sr_hip.h) against the per-layer route that rounds 1-3 ran (the one pinned by goldens G-g / G-n): same launches, so
every output, gradient and BatchNorm buffer must be BIT-identical; and the forward-reuse rule of the ESRGAN step
(esrgan_model.py:38-39,65-72: net_d(gt) twice, net_d(output) three times on unchanged weights) — five calls through two
kept forwards leave the same logits, gradients and running statistics as five real forwards."""
import copy

import numpy as np
import pytest
import torch

import image_restoration_amd as ira
from image_restoration_amd.utils import synth

pytestmark = pytest.mark.gpu


def _net(dev, kind='VGGStyleDiscriminator128', nf=16, dtype='fp32', seed=61):
    size = 128 if kind.endswith('128') else 256
    net = ira.build_network(dict(type=kind, num_in_ch=3, num_feat=nf, compute_dtype=dtype)).to(dev)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth.vgg128_state_dict(seed, 3, nf, size).items()}, strict=True)
    return net.train()


def _run(net, x, weight, route, need_x=True):
    net.zero_grad(set_to_none=True)
    xx = x.clone().requires_grad_(need_x)
    out = net(xx) if route == 'driver' else net.forward_layers(xx)
    (out * weight).sum().backward()
    return out.detach().clone(), (xx.grad.clone() if need_x else None), {n: p.grad.clone() for n, p in net.named_parameters()}


@pytest.mark.parametrize('kind,nf,n', [('VGGStyleDiscriminator128', 16, 3), ('VGGStyleDiscriminator128', 64, 2),
                                      ('VGGStyleDiscriminator256', 16, 2)])
@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_driver_is_bit_identical_to_the_per_layer_route(cuda, kind, nf, n, dtype):
    size = 128 if kind.endswith('128') else 256
    a = _net(cuda, kind, nf, dtype)
    b = copy.deepcopy(a)
    x = torch.from_numpy(synth.uniform_input(5, (n, 3, size, size))).to(cuda)
    wgt = torch.from_numpy(synth.uniform_input(6, (n, 1))).to(cuda) - 0.5
    for _ in range(2):   # twice: the second pass starts from moved running statistics
        oa, gxa, gpa = _run(a, x, wgt, 'driver')
        ob, gxb, gpb = _run(b, x, wgt, 'layers')
        assert torch.equal(oa, ob)
        assert torch.equal(gxa, gxb)
        for k in gpa:
            assert torch.equal(gpa[k], gpb[k]), k
        for (k, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
            assert torch.equal(ba, bb), k
    # a frozen discriminator (generator phase): input gradient only, no parameter gradient appears
    for net in (a, b):
        for p in net.parameters():
            p.requires_grad = False
    oa, gxa, gpa = _run_frozen(a, x, wgt, 'driver')
    ob, gxb, gpb = _run_frozen(b, x, wgt, 'layers')
    assert torch.equal(oa, ob) and torch.equal(gxa, gxb)
    # eval mode reads the running statistics and moves nothing
    a.eval(), b.eval()
    before = {k: v.clone() for k, v in a.named_buffers()}
    with torch.no_grad():
        assert torch.equal(a(x), b.forward_layers(x))
    for k, v in a.named_buffers():
        assert torch.equal(v, before[k]), k


def _run_frozen(net, x, weight, route):
    xx = x.clone().requires_grad_(True)
    out = net(xx) if route == 'driver' else net.forward_layers(xx)
    (out * weight).sum().backward()
    assert all(p.grad is None or not p.requires_grad for p in net.parameters())
    return out.detach().clone(), xx.grad.clone(), None


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_five_calls_through_two_kept_forwards_equal_five_forwards(cuda, dtype):
    """The ESRGAN step's discriminator calls in the reference's order (gt, out, out, gt, out) with the three backward passes:
    once as five real forwards, once with the repeats served by the kept forwards.  Logits, input gradient, accumulated parameter
    gradients, running statistics and num_batches_tracked are bit-identical."""
    nf, n = 16, 4
    a = _net(cuda, nf=nf, dtype=dtype)
    b = copy.deepcopy(a)
    gt = torch.from_numpy(synth.uniform_input(7, (n, 3, 128, 128))).to(cuda)
    out = torch.from_numpy(synth.uniform_input(8, (n, 3, 128, 128))).to(cuda).requires_grad_(True)
    w = [torch.from_numpy(synth.uniform_input(20 + i, (n, 1))).to(cuda) - 0.5 for i in range(3)]

    def step(net, reuse):
        kept = {}

        def call(x, tag):
            slot = []
            y = net(x, kept=kept.get(tag) if reuse else None, slot=slot)
            kept[tag] = slot[0]
            return y
        res = {}
        for p in net.parameters():
            p.requires_grad = False
        with torch.no_grad():
            res['real_const'] = call(gt, 'gt')
        f2 = call(out, 'out')
        out.grad = None
        (f2 * w[0]).sum().backward()
        res['f2'], res['dx'] = f2.detach().clone(), out.grad.clone()
        for p in net.parameters():
            p.requires_grad = True
        net.zero_grad(set_to_none=True)
        fake = out.detach()
        with torch.no_grad():
            res['fake_const'] = call(fake, 'out')
        f4 = call(gt, 'gt')
        (f4 * w[1]).sum().backward()
        f5 = call(fake, 'out')
        (f5 * w[2]).sum().backward()
        res['f4'], res['f5'] = f4.detach().clone(), f5.detach().clone()
        res.update({'g_' + k: p.grad.clone() for k, p in net.named_parameters()})
        res.update({'b_' + k: v.clone() for k, v in net.named_buffers()})
        return res
    for it in range(2):
        ra, rb = step(a, False), step(b, True)
        for k in ra:
            assert torch.equal(ra[k], rb[k]), (it, k)
    assert int(a.bn0_1.num_batches_tracked) == 10 == int(b.bn0_1.num_batches_tracked)
    assert torch.equal(ra['real_const'], ra['f4']) and torch.equal(ra['f2'], ra['f5']) and torch.equal(ra['f2'], ra['fake_const'])


def test_a_kept_forward_is_refused_when_input_or_weights_moved(cuda):
    net = _net(cuda, nf=16)
    x = torch.from_numpy(synth.uniform_input(9, (2, 3, 128, 128))).to(cuda)
    slot = []
    y0 = net(x, slot=slot)
    kept = slot[0]
    x2 = x.clone()
    x2.mul_(0.5)
    slot = []
    y1 = net(x2, kept=kept, slot=slot)    # other storage: a real forward runs
    assert slot[0] is not kept and not torch.equal(y0, y1)
    x.add_(0.25)                          # same storage, version moved
    slot = []
    net(x, kept=kept, slot=slot)
    assert slot[0] is not kept
    slot = []
    net(x, slot=slot)
    kept = slot[0]
    with torch.no_grad():
        net.linear2.bias.add_(1.0)        # weights moved
    slot = []
    y2 = net(x, kept=kept, slot=slot)
    assert slot[0] is not kept
    net.invalidate_packed()               # the raw-pointer writers' signal (FlatAdam.step)
    slot2 = []
    net(x, kept=slot[0], slot=slot2)
    assert slot2[0] is not slot[0]
    assert torch.isfinite(y2).all()

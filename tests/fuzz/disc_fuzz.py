"""Randomised check of the discriminators on the HIP path (fp32) against their PyTorch-CPU oracles: UNetDiscriminatorSN over random
widths / sizes / skip settings / train-eval mode (oracle/unet_discriminator_ref.py), VGGStyleDiscriminator128 / 256 over random widths
and batch sizes (oracle/discriminator_ref.py): logits, input gradient and every parameter gradient.  Exit code 1 on a mismatch."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import image_restoration_amd as ira
from image_restoration_amd.utils import synth
from oracle import discriminator_ref as D
from oracle.unet_discriminator_ref import UNetDiscriminatorSNRef

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device('cuda:0')
bad = 0


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).norm() / (b.norm() + 1e-30))


for it in range(N):
    nf, cin, skip, train = random.choice([8, 16, 24, 32]), random.choice([1, 3]), random.random() < 0.8, random.random() < 0.7
    n, h, w = random.choice([1, 2, 3]), 8 * random.choice([1, 2, 3, 5, 8]), 8 * random.choice([1, 2, 4, 7])
    torch.manual_seed(it)
    ref = UNetDiscriminatorSNRef(cin, nf, skip)
    net = ira.build_network(dict(type='UNetDiscriminatorSN', num_in_ch=cin, num_feat=nf, skip_connection=skip))
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.to(dev)
    with torch.no_grad():  # let the power iterations settle on both sides (raw u, v give weights of norm 1e2 .. 1e4 and 1e10 logits)
        for _ in range(10):
            warm = torch.rand(1, cin, 16, 16)
            ref.train()(warm), net.train()(warm.to(dev))
    ref.train(train), net.train(train)
    x = torch.rand(n, cin, h, w)
    R = torch.randn(n, 1, h, w)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    (yr * R).sum().backward()
    xc = x.to(dev).requires_grad_(True)
    y = net(xc)
    (y * R.to(dev)).sum().backward()
    gr = dict(ref.named_parameters())
    e = max([rel(y, yr), rel(xc.grad, xr.grad)] + [rel(p.grad, gr[k].grad) for k, p in net.named_parameters()])
    ok = e < 2e-3
    bad += not ok
    print(f'unet {it:2d} nf={nf} cin={cin} skip={int(skip)} train={int(train)} x={n}x{h}x{w}: worst {e:.1e} {"ok" if ok else "MISMATCH"}', flush=True)

for it in range(max(N // 2, 4)):
    size = random.choice([128, 128, 256])
    nf, n = random.choice([4, 8, 12]), random.choice([2, 3, 5])
    train = random.random() < 0.7
    sd_np = synth.vgg128_state_dict(300 + it, 3, nf, size)
    net = ira.build_network(dict(type=f'VGGStyleDiscriminator{size}', num_in_ch=3, num_feat=nf)).to(dev)
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}, strict=True)
    net.train(train)
    sd = {k: torch.from_numpy(np.asarray(v)).clone() for k, v in sd_np.items()}
    for k in sd:
        if sd[k].is_floating_point() and 'running' not in k:
            sd[k].requires_grad_(True)
    x = torch.rand(n, 3, size, size)
    R = torch.randn(n, 1)
    xr = x.clone().requires_grad_(True)
    yr = D.vgg128_forward(xr, sd, train=train, input_size=size)
    (yr * R).sum().backward()
    xc = x.to(dev).requires_grad_(True)
    y = net(xc)
    (y * R.to(dev)).sum().backward()
    e = max([rel(y, yr), rel(xc.grad, xr.grad)] + [rel(p.grad, sd[k].grad) for k, p in net.named_parameters()] +
            [rel(b, sd[k]) for k, b in net.named_buffers() if 'running' in k])
    ok = e < 2e-2  # 10 BatchNorm + LeakyReLU stages: fp32 gradients of this net carry 1e-3 noise on CPU as well (tests/test_training_gpu.py)
    bad += not ok
    print(f'vgg{size} {it:2d} nf={nf} n={n} train={int(train)}: worst {e:.1e} {"ok" if ok else "MISMATCH"}', flush=True)
print('mismatches:', bad)
sys.exit(1 if bad else 0)

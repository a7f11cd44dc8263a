"""Robustness sweep: generator and discriminators, forward + backward, over batch sizes and resolutions the tests do not pin,
in fp32 and bf16.  Prints every (network, dtype, shape) that raises or produces non-finite values; exit code 1 if any."""
import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import image_restoration_amd as ira

dev = torch.device('cuda:0')
bad = []


def run(name, net, shape, dtype):
    try:
        x = torch.rand(*shape, device=dev, requires_grad=True)
        y = net(x)
        y.float().mean().backward()
        ok = bool(torch.isfinite(y).all()) and bool(torch.isfinite(x.grad).all()) and all(
            p.grad is None or bool(torch.isfinite(p.grad).all()) for p in net.parameters())
        if not ok:
            bad.append((name, dtype, shape, 'non-finite'))
        for p in net.parameters():
            p.grad = None
    except Exception as e:  # noqa: BLE001
        bad.append((name, dtype, shape, repr(e)[:300]))
        traceback.print_exc(limit=1)
    print(name, dtype, shape, 'ok' if not bad or bad[-1][:3] != (name, dtype, shape) else 'FAILED', flush=True)


for dtype in ('fp32', 'bf16'):
    g = ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=2, num_grow_ch=32,
                               compute_dtype=dtype)).to(dev).train()
    for shape in ((1, 3, 17, 23), (3, 3, 40, 40), (5, 3, 64, 48), (48, 3, 32, 32), (64, 3, 32, 32), (2, 3, 200, 136), (1, 3, 256, 256),
                  (24, 3, 128, 128)):
        run('RRDBNet', g, shape, dtype)
    del g
    g2 = ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=2, num_feat=32, num_block=1, num_grow_ch=16,
                                compute_dtype=dtype)).to(dev).train()
    for shape in ((2, 3, 34, 50), (16, 3, 64, 64)):
        run('RRDBNet x2', g2, shape, dtype)
    del g2
    vgg = ira.build_network(dict(type='VGGStyleDiscriminator128', num_in_ch=3, num_feat=64, compute_dtype=dtype)).to(dev).train()
    for n in (2, 5, 48, 64, 96):
        run('VGG128', vgg, (n, 3, 128, 128), dtype)
    del vgg
    unet = ira.build_network(dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=64, compute_dtype=dtype)).to(dev).train()
    for shape in ((1, 3, 64, 64), (3, 3, 96, 160), (6, 3, 256, 256), (40, 3, 128, 128), (2, 3, 512, 512), (64, 3, 64, 64), (1, 3, 1024, 1024)):
        run('UNetSN', unet, shape, dtype)
    del unet
    torch.cuda.empty_cache()
print('failures:', len(bad))
for b in bad:
    print('  ', b)
sys.exit(1 if bad else 0)

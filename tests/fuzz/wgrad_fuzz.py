"""Randomised differential check of the weight-gradient entry points: sr_conv3x3_wgrad_f32 against torch's CPU autograd
(float64), sr_conv3x3_wgrad_bf16 against the same reference on bf16-rounded operands.  Exit code 1 on any mismatch."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from image_restoration_amd import hip_ops as ops

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for it in range(N):
    n = random.choice([1, 2, 3, 5, 8, 17, 33])
    cin = random.choice([8, 16, 32, 48, 64, 96, 128, 160, 192, 256, 512])
    cout = random.choice([3, 8, 16, 32, 40, 64, 96, 128, 256])
    h, w = random.choice([1, 2, 4, 7, 8, 16, 17, 32, 33, 64, 100]), random.choice([1, 4, 8, 16, 31, 32, 33, 64, 65, 128, 130])
    if n * cin * h * w > 3e6 or n * cout * h * w > 3e6 or cin * cout * n * h * w > 6e9:
        continue
    ups = random.random() < 0.15 and n * cout * h * w * 4 < 2e6
    g = torch.Generator().manual_seed(1000 + it)
    x = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16).float()
    H, W = (2 * h, 2 * w) if ups else (h, w)
    dy = torch.randn(n, cout, H, W, generator=g).to(torch.bfloat16).float()
    xin = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    ref_w = torch.nn.grad.conv2d_weight(xin.double(), (cout, cin, 3, 3), dy.double(), padding=1)
    ref_b = dy.double().sum(dim=(0, 2, 3))
    xc, dc = x.cuda(), dy.cuda()
    dw32, db32 = ops.conv3x3_wgrad(ops.nchw_to_cb8(xc), ops.nchw_to_cb8(dc), cout, cin, upsample=ups)
    cin16 = (cin + 15) // 16 * 16
    src16 = ops.nchw_to_cb16(F.pad(xc, (0, 0, 0, 0, 0, cin16 - cin)))
    dw16, db16 = ops.conv3x3_wgrad_bf16(src16, ops.nchw_to_cb16(dc), cout, cin, upsample=ups)
    sw, sb = float(ref_w.abs().max()) + 1e-9, float(ref_b.abs().max()) + 1e-9
    e = [float((dw32.cpu().double() - ref_w).abs().max()) / sw, float((db32.cpu().double() - ref_b).abs().max()) / sb,
         float((dw16.cpu().double() - ref_w).abs().max()) / sw, float((db16.cpu().double() - ref_b).abs().max()) / sb]
    ok = max(e) < 3e-4
    bad += not ok
    print(f'{it:3d} n={n} cin={cin} cout={cout} {h}x{w} ups={int(ups)}: ' + ' '.join(f'{v:.1e}' for v in e) + (' ok' if ok else ' MISMATCH'), flush=True)
print('mismatches:', bad)
sys.exit(1 if bad else 0)

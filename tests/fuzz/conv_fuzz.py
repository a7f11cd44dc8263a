"""Randomised differential check of the single-conv entry points over shapes the fixed tests do not list:
sr_conv3x3_f32 against torch's CPU convolution, sr_conv3x3_bf16 against sr_conv3x3_f32 (bf16-rounded operands), with
upsampling, residuals, the LeakyReLU-backward mask and concat-style channel counts.  Exit code 1 on any mismatch."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from image_restoration_amd import hip_ops as ops

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bad = 0
for it in range(N):
    n = random.choice([1, 1, 2, 3, 5, 8, 17])
    cin = random.choice([8, 16, 24, 32, 48, 64, 96, 128, 160, 192, 256, 320])
    cout = random.choice([3, 8, 16, 32, 40, 64, 96, 128, 192])
    h, w = random.choice([1, 2, 3, 7, 8, 15, 16, 17, 31, 32, 33, 48, 64, 65, 96, 128]), random.choice([1, 5, 8, 16, 31, 32, 33, 40, 64, 70, 128, 200])
    if n * cin * h * w > 6e6 or n * cout * h * w > 6e6:
        continue
    ups = random.random() < 0.2 and h * w * n * cout * 4 < 4e6
    slope = random.choice([1.0, 0.2, 0.0])
    res = random.random() < 0.3
    g = torch.Generator().manual_seed(it)
    x = torch.randn(n, cin, h, w, generator=g).to(torch.bfloat16).float()
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(torch.bfloat16).float()
    b = torch.randn(cout, generator=g) * 0.1
    H, W = (2 * h, 2 * w) if ups else (h, w)
    r1 = torch.randn(n, cout, H, W, generator=g).to(torch.bfloat16).float() if res else None
    xin = F.interpolate(x, scale_factor=2, mode='nearest') if ups else x
    ref = F.conv2d(xin.double(), wt.double(), b.double(), padding=1)
    ref = torch.where(ref > 0, ref, ref * slope) * 0.7
    if res:
        ref = ref + 0.3 * r1.double()
    xc, wc, bc = x.cuda(), wt.cuda(), b.cuda()
    o32 = ops.conv3x3(ops.nchw_to_cb8(xc), ops.PackedConv(wc, bc), upsample=ups, act_slope=slope, alpha=0.7,
                      res1=ops.nchw_to_cb8(r1.cuda()) if res else None, beta1=0.3)
    y32 = ops.cb8_to_nchw(o32, cout).cpu().double()
    o16 = ops.conv3x3_bf16(ops.nchw_to_cb16(xc), ops.PackedConvBF16(wc, bc), upsample=ups, act_slope=slope, alpha=0.7,
                           res1=ops.nchw_to_cb16(r1.cuda()) if res else None, beta1=0.3)
    y16 = ops.cb16_to_nchw(o16, cout).cpu().double()
    scale = float(ref.abs().max()) + 1e-6
    e32 = float((y32 - ref).abs().max()) / scale
    e16 = float(((y16 - ref).abs() - ref.abs() * 2 ** -8).clamp_min(0).max()) / scale
    ok = e32 < 2e-5 and e16 < 2e-4
    bad += not ok
    print(f'{it:3d} n={n} cin={cin} cout={cout} {h}x{w} ups={int(ups)} slope={slope} res={int(res)}: f32 {e32:.1e} bf16 {e16:.1e} {"ok" if ok else "MISMATCH"}',
          flush=True)
print('mismatches:', bad)
sys.exit(1 if bad else 0)

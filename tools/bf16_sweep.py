"""Per-shape timing of sr_conv3x3_bf16 (development aid): time vs Cin / batch to separate fixed from per-byte cost."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_restoration_amd import hip_ops as ops


def bench(n, cin, cout, h, w, ups=False, res=False, iters=50):
    g = torch.Generator().manual_seed(0)
    src = ops.CB16(torch.randn(n, cin // 16, h, w, 16, generator=g).to(torch.bfloat16).cuda())
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * 0.02).cuda()
    b = torch.zeros(cout).cuda()
    pc = ops.PackedConvBF16(wt, b)
    H, W = (2 * h, 2 * w) if ups else (h, w)
    out = ops.CB16.zeros(n, cout, H, W, 'cuda')
    r1 = ops.CB16.zeros(n, cout, H, W, 'cuda') if res else None
    for _ in range(5):
        ops.conv3x3_bf16(src, pc, out=out, upsample=ups, act_slope=0.2, res1=r1, beta1=0.2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv3x3_bf16(src, pc, out=out, upsample=ups, act_slope=0.2, res1=r1, beta1=0.2)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * 9 * cin * cout * n * H * W
    by = 2.0 * (n * h * w * cin + n * H * W * cout * (2 if res else 1))
    print(f'n={n:3d} cin={cin:3d} cout={cout:2d} {h}x{w} ups={int(ups)} res={int(res)}: {us:8.1f} us  {fl / us / 1e6:7.1f} TF  '
          f'{by / us / 1e3:7.1f} GB/s', flush=True)


if __name__ == '__main__':
    for cin in (64, 96, 128, 160):
        bench(16, cin, 32, 128, 128)
    bench(16, 192, 64, 128, 128, res=True)
    bench(16, 64, 64, 128, 128)
    for n in (4, 64):
        bench(n, 64, 32, 128, 128)
        bench(n, 160, 32, 128, 128)
        bench(n, 192, 64, 128, 128, res=True)
    bench(16, 64, 64, 128, 128, ups=True)
    bench(16, 64, 64, 256, 256, ups=True)
    bench(16, 64, 64, 512, 512)

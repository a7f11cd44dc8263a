"""Development aid: the large few-channel convs on the per-tile kernel vs conv_stream_bf16_kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_restoration_amd import _lib
from image_restoration_amd import hip_ops as H

dev = torch.device('cuda'); lib = _lib.load()
for (cin, n, h, w, mask, up) in [(64, 32, 512, 512, False, False), (64, 32, 512, 512, True, False), (16, 32, 512, 512, False, False), (64, 16, 512, 512, False, False),
                                 (64, 16, 256, 256, False, True), (64, 32, 256, 256, False, False)]:
    g = torch.Generator().manual_seed(1)
    x = H.CB16(torch.randn(n, cin // 16, h, w, 16, generator=g).to(torch.bfloat16).to(dev))
    wt = (torch.randn(64, cin, 3, 3, generator=g) * 0.05).to(dev)
    pc = H.PackedConvBF16(wt, torch.zeros(64, device=dev))
    oh, ow = (2 * h, 2 * w) if up else (h, w)
    out = H.CB16.empty(n, 64, oh, ow, dev)
    kw = dict(act_slope=0.2, upsample=up)
    if mask:
        kw = dict(act_slope=1.0, mask=H.CB16(torch.randn(n, 4, oh, ow, 16, generator=g).to(torch.bfloat16).to(dev)))
    res = []
    for s in (0, 1):
        lib.sr_dev_set_conv_stream(s)
        for _ in range(3):
            H.conv3x3_bf16(x, pc, out, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            H.conv3x3_bf16(x, pc, out, **kw)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 100)
    by = 2.0 * (n * h * w * cin + n * oh * ow * 64 * (2 if mask else 1))
    print(f'cin {cin} n {n} {h}x{w} mask {mask} up {up}: per-tile {res[0]:.1f} us ({by / res[0] / 1e6:.2f} TB/s)  stream {res[1]:.1f} us ({by / res[1] / 1e6:.2f} TB/s)  x{res[0] / res[1]:.2f}', flush=True)

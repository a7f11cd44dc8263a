"""Development aid: where the fused dense-block kernel's x1 differs from the conv-by-conv result (per channel / row / column)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from image_restoration_amd import _lib
from image_restoration_amd import hip_ops as H
from test_chain_bf16_gpu import _fresh, _rdb, _steps
dev = torch.device('cuda'); lib = _lib.load()
n, h, w, nf, gc = 8, 128, 128, 64, 32
packs = _rdb(dev, nf, gc, 3)
cat_a, nxt_a = _fresh(dev, n, nf, gc, h, w, 5)
lib.sr_set_conv_chain(0)
for src, pc, out, kw in _steps(cat_a, nxt_a, packs, nf, gc):
    H.conv3x3_bf16(src, pc, out, **kw)
lib.sr_set_conv_chain(3)
cat_b, nxt_b = _fresh(dev, n, nf, gc, h, w, 5)
H.conv3x3_chain_bf16(_steps(cat_b, nxt_b, packs, nf, gc), None, 0)
torch.cuda.synchronize()
a = cat_a.buf[:, 4].float().cpu(); b = cat_b.buf[:, 4].float().cpu()   # [n, h, w, 16]
bad = (a != b)
print('block 4: bad fraction', float(bad.float().mean()))
print('by channel', [round(float(bad[..., c].float().mean()), 3) for c in range(16)])
print('by row mod 16', [round(float(bad[:, r::16].float().mean()), 3) for r in range(16)])
print('by col mod 32', [round(float(bad[:, :, c::32].float().mean()), 2) for c in range(32)])
idx = bad.nonzero()[:12]
for i in idx:
    i = tuple(int(v) for v in i)
    print(i, 'expected', float(a[i]), 'got', float(b[i]), 'block5 same pos exp', float(cat_a.buf[i[0], 5, i[1], i[2], i[3]]), 'got', float(cat_b.buf[i[0], 5, i[1], i[2], i[3]]))

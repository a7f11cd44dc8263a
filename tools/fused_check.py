"""Development aid: the fused dense-block kernel (sr_set_conv_chain(3)) against conv-by-conv launches, bit for bit, then timings."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from image_restoration_amd import _lib
from image_restoration_amd import hip_ops as H
from test_chain_bf16_gpu import _fresh, _rdb, _steps

dev = torch.device('cuda')
lib = _lib.load()
ok = True
MODE = int(os.environ.get('FUSED_MODE', '3'))
CFGS = [(8, 128, 128), (16, 128, 128), (3, 64, 64), (12, 160, 100), (20, 128, 128), (24, 128, 128), (20, 128, 128)]
if os.environ.get('STRESS'):
    CFGS = [(16, 128, 128), (20, 128, 128), (12, 160, 100), (4, 544, 544), (3, 320, 1088)] * 3
for (n, h, w) in CFGS:
    nf, gc = 64, 32
    packs = _rdb(dev, nf, gc, 3)
    cat_a, nxt_a = _fresh(dev, n, nf, gc, h, w, 5)
    lib.sr_set_conv_chain(0)
    for src, pc, out, kw in _steps(cat_a, nxt_a, packs, nf, gc):
        H.conv3x3_bf16(src, pc, out, **kw)
    lib.sr_set_conv_chain(MODE)
    sync = None
    for rep in range(4):
        cat_b, nxt_b = _fresh(dev, n, nf, gc, h, w, 5)
        _, sync = H.conv3x3_chain_bf16(_steps(cat_b, nxt_b, packs, nf, gc), sync, call_index=rep)
        torch.cuda.synchronize()
        ab = int(sync[0])
        eq_cat = torch.equal(cat_a.buf, cat_b.buf)
        eq_out = torch.equal(nxt_a.buf[:, :nf // 16], nxt_b.buf[:, :nf // 16])
        msg = ''
        if not eq_cat:
            d = (cat_a.buf.float() - cat_b.buf.float()).abs()
            per_block = [float(d[:, b].max()) for b in range(d.shape[1])]
            bad = (cat_a.buf != cat_b.buf) & ~(torch.isnan(cat_a.buf) & torch.isnan(cat_b.buf))
            idx = bad.nonzero()
            imgs = sorted(set(idx[:, 0].tolist()))
            tiles = sorted(set((int(i[0]), int(i[2]) // 16, int(i[3]) // 32) for i in idx[:4000]))[:12]
            msg = f' cat bad blocks {[b for b in range(d.shape[1]) if not per_block[b] == 0.0]} images {imgs} tiles(img,ty,tx) {tiles}'
        if not eq_out:
            d = (nxt_a.buf[:, :4].float() - nxt_b.buf[:, :4].float()).abs()
            msg += f' out max diff {float(d.max()):.4g} frac {float((d > 0).float().mean()):.4f}'
        print(f'n={n} {h}x{w} rep {rep}: abort={ab} cat_equal={eq_cat} out_equal={eq_out}{msg}', flush=True)
        ok = ok and eq_cat and eq_out and ab == 0
print('ALL EQUAL' if ok else 'MISMATCH', flush=True)
if ok and not os.environ.get('NO_TIME'):
    import chain_bench as B
    for mode in (2, 3):
        lib.sr_set_conv_chain(mode)
        print('mode', mode, flush=True)
        B.bench(16, 128, 128)
        B.bench(32, 128, 128)
        B.bench(8, 128, 128)

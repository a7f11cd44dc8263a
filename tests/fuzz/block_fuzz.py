"""Randomised differential check of the one-launch dense blocks (sr_conv3x3_chain_bf16: fused dense-block kernel where eligible,
persistent chain otherwise) and of the streaming conv against conv-by-conv launches of the per-tile kernel: random batch sizes,
heights (multiples of 16, and a few that are not: fallback), widths that are no multiple of the 32-pixel tile, forward and transposed
blocks, one or two residual sources, consecutive calls on one sync block; the fused kernel on eight waves of two rows or four waves of
four (sr_dev_set_fused_wave4), and with the caller's word that x1..x4 are scratch (sr_dev_set_chain_mids_scratch: only the ring a
neighbouring tile reads is stored — the OUTPUT is compared then).  Everything accumulates in the same order, so the comparison is
BIT for bit.  Exit code 1 on any mismatch."""
import sys, os, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from image_restoration_amd import _lib
from image_restoration_amd import hip_ops as H

random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 24
dev = torch.device('cuda')
lib = _lib.load()
nf, gc = 64, 32
bad = 0
for it in range(N):
    h = random.choice([16, 32, 48, 64, 80, 96, 128, 160, 208, 272, 24])
    w = random.choice([8, 31, 32, 33, 64, 70, 96, 100, 128, 130, 200, 260])
    n = random.choice([1, 2, 3, 5, 9, 16, 23])
    while n * h * w > 700000:
        n = max(1, n // 2)
    transposed = random.random() < 0.4
    two_res = random.random() < 0.5
    mode = random.choice([2, 3, 3, 3])
    wave4 = random.random() < 0.4
    scratch = random.random() < 0.4
    g = torch.Generator().manual_seed(1000 + it)
    packs = []
    for k in range(1, 6):
        cout, cin = (nf if k == 5 else gc), nf + (k - 1) * gc
        wt = (torch.randn(cout, cin, 3, 3, generator=g) * (0.6 / (cin * 9) ** 0.5)).to(dev)
        b = None if transposed else (torch.randn(cout, generator=g) * 0.05).to(dev)
        packs.append(H.PackedConvBF16(wt, b, first_seg=nf, seg=gc))
    fwd = H.CB16(torch.randn(n, (nf + 4 * gc) // 16, h, w, 16, generator=g).to(torch.bfloat16).to(dev))
    extra = H.CB16(torch.randn(n, nf // 16, h, w, 16, generator=g).to(torch.bfloat16).to(dev))
    x0 = torch.randn(n, nf // 16, h, w, 16, generator=g).to(torch.bfloat16)

    def fresh():
        buf = torch.full((n, (nf + 4 * gc) // 16, h, w, 16), 7.0, dtype=torch.bfloat16)
        buf[:, :nf // 16] = x0
        return H.CB16(buf.to(dev)), H.CB16(torch.full((n, (nf + 4 * gc) // 16, h, w, 16), -3.0, dtype=torch.bfloat16, device=dev))

    def steps(cat, nxt):
        st = []
        for k in range(1, 5):
            kw = dict(mask=fwd.slice(nf + (4 - k) * gc, gc), mask_slope=0.2) if transposed else dict(act_slope=0.2)
            st.append((cat.slice(0, nf + (k - 1) * gc), packs[k - 1], cat.slice(nf + (k - 1) * gc, gc), kw))
        kw = dict(alpha=1.0 if transposed else 0.2, res1=cat.slice(0, nf), beta1=0.2 if two_res else 1.0)
        if two_res:
            kw.update(res2=extra, beta2=1.0)
        st.append((cat, packs[4], nxt.slice(0, nf), kw))
        return st

    lib.sr_set_conv_chain(0)
    lib.sr_dev_set_conv_stream(0)
    cat_a, nxt_a = fresh()
    for src, pc, out, kw in steps(cat_a, nxt_a):
        H.conv3x3_bf16(src, pc, out, **kw)
    lib.sr_set_conv_chain(mode)
    lib.sr_dev_set_conv_stream(1)
    lib.sr_dev_set_fused_wave4(int(wave4))
    lib.sr_dev_set_chain_mids_scratch(int(scratch))
    ok, sync = True, None
    for rep in range(2):
        cat_b, nxt_b = fresh()
        _, sync = H.conv3x3_chain_bf16(steps(cat_b, nxt_b), sync, call_index=rep)
        torch.cuda.synchronize()
        ok = ok and int(sync[0]) == 0 and torch.equal(nxt_a.buf[:, :nf // 16], nxt_b.buf[:, :nf // 16])
        ok = ok and torch.equal(cat_a.buf[:, :nf // 16], cat_b.buf[:, :nf // 16])
        if not (scratch and not transposed):   # (a forward block whose intermediates are scratch leaves their insides unwritten)
            ok = ok and torch.equal(cat_a.buf, cat_b.buf)
    lib.sr_dev_set_fused_wave4(0)
    lib.sr_dev_set_chain_mids_scratch(0)
    bad += not ok
    print(f'{it:3d} n={n} {h}x{w} mode={mode} transposed={int(transposed)} two_res={int(two_res)} wave4={int(wave4)} scratch={int(scratch)}: '
          f'{"ok" if ok else "MISMATCH"}', flush=True)
lib.sr_set_conv_chain(3)
print('mismatches:', bad)
sys.exit(1 if bad else 0)

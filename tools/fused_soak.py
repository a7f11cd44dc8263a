"""Development aid: the fused dense block launched over and over on the bench's shapes (a rare hand-off / LDS ordering slip shows up as
one tile in thousands): every launch's output is compared on the device with the conv-by-conv result, bit for bit.
usage: python tools/fused_soak.py [launches per case]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from image_restoration_amd import _lib, hip_ops as H
from test_chain_bf16_gpu import _fresh, _rdb, _steps

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
lib = _lib.load()
lib.sr_dev_set_fused_wave4.argtypes = [C.c_int]
lib.sr_dev_set_chain_mids_scratch.argtypes = [C.c_int]
dev = torch.device('cuda')
nf, gc = 64, 32
bad_total = 0
for n, h, w in ((16, 128, 128), (32, 128, 128), (4, 544, 544), (32, 32, 32)):
    packs = _rdb(dev, nf, gc, 3)
    lib.sr_set_conv_chain(0)
    cat_a, nxt_a = _fresh(dev, n, nf, gc, h, w, 5)
    for src, pc, out, kw in _steps(cat_a, nxt_a, packs, nf, gc):
        H.conv3x3_bf16(src, pc, out, **kw)
    ref = nxt_a.buf[:, :nf // 16].clone()
    lib.sr_set_conv_chain(3)
    for wave4, scratch in ((0, 0), (0, 1), (1, 0), (1, 1)):
        lib.sr_dev_set_fused_wave4(wave4)
        lib.sr_dev_set_chain_mids_scratch(scratch)
        cat_b, nxt_b = _fresh(dev, n, nf, gc, h, w, 5)
        steps = _steps(cat_b, nxt_b, packs, nf, gc)
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        sync = None
        for it in range(reps):
            if it % 200 == 0:
                sync = None   # (a sync block serves 256 calls: every call_index has its own ticket counter, zero before its launch)
            _, sync = H.conv3x3_chain_bf16(steps, sync, call_index=it % 200)
            bad += (nxt_b.buf[:, :nf // 16] != ref).any()
            nxt_b.buf[:, :nf // 16].fill_(-3.0)
            if not scratch:
                cat_b.buf[:, nf // 16:].fill_(7.0)
        torch.cuda.synchronize()
        print(f'n={n} {h}x{w} waves={"4" if wave4 else "8"} ring-only stores={scratch}: {reps} launches, {int(bad)} differ, abort word {int(sync[0])}', flush=True)
        bad_total += int(bad) + int(sync[0])
lib.sr_dev_set_fused_wave4(0)
lib.sr_dev_set_chain_mids_scratch(0)
sys.exit(1 if bad_total else 0)

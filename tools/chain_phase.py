"""Development aid: where a work item of the persistent conv chain spends its time (clock ticks of s_memtime, 100 MHz)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from image_restoration_amd import _lib, hip_ops as H
from test_chain_bf16_gpu import _fresh, _rdb, _steps


def run(n, h, w, nf=64, gc=32):
    lib = _lib.load()
    lib.sr_dev_chain_phase_clocks.argtypes = [C.c_void_p]
    dev = torch.device('cuda')
    packs = _rdb(dev, nf, gc, 3)
    cat, nxt = _fresh(dev, n, nf, gc, h, w, 5)
    steps = _steps(cat, nxt, packs, nf, gc)
    ntiles = n * (h // 32) * ((w + 31) // 32)
    dbg = torch.zeros(5 * ntiles * 16, dtype=torch.int64, device=dev)
    for it in range(3):
        lib.sr_dev_chain_phase_clocks(dbg.data_ptr() if it == 2 else None)
        H.conv3x3_chain_bf16(steps, None, 0)
    lib.sr_dev_chain_phase_clocks(None)
    torch.cuda.synchronize()
    t = dbg.cpu().view(5, ntiles, 16).double()
    t0 = t[:, :, 0].min()
    print(f'n={n} {h}x{w}: {ntiles} tiles; ticks are s_memtime units')
    for k in range(5):
        u = t[k]
        end = u[:, 0] + u[:, 1:6].sum(1)
        print(f' conv{k + 1}: start {float(u[:, 0].min() - t0):8.0f}..{float(u[:, 0].max() - t0):8.0f} | claim {float(u[:, 1].mean()):6.0f} | wait {float(u[:, 2].mean()):7.0f} '
              f'(max {float(u[:, 2].max()):7.0f}) | acquire {float(u[:, 3].mean()):6.0f} | tile {float(u[:, 4].mean()):7.0f} | drain+flag {float(u[:, 5].mean()):6.0f} '
              f'|| setup {float(u[:, 8].mean()):5.0f} | issue {float(u[:, 9].mean()):5.0f} | first data {float(u[:, 10].mean()):6.0f} | loop {float(u[:, 11].mean()):7.0f} '
              f'| epilogue {float(u[:, 12].mean()):6.0f}')
    # per-XCD timeline: the cycle counters of different XCDs are not aligned
    xcd = t[:, :, 7].long() & 7
    for x in range(8):
        sel = xcd == x
        if sel.any():
            st = t[:, :, 0][sel]
            en = (t[:, :, 0] + t[:, :, 1:6].sum(2))[sel]
            print(f'  XCD {x}: items {int(sel.sum())}  span {float(en.max() - st.min()):9.0f} cycles')


if __name__ == '__main__':
    run(16, 128, 128)
    run(32, 128, 128)

"""Development aid: stage timeline of the fused dense-block kernel (clock ticks of s_memtime, 100 MHz = 10 ns).

Stamps per workgroup (sr_dev_fused_phase_clocks): 0 round start; per input group s: 4s+2 first step runs (its operands landed), 4s+3 conv s+1
published; per fetched tile t = s+1: 4t flag check starts, 4t+1 flags seen; 24 end of the block."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from image_restoration_amd import _lib, hip_ops as H
from test_chain_bf16_gpu import _fresh, _rdb, _steps

NAMES = {0: 'start', 1: 'step 0 runs'}
for _t in range(1, 5):
    _b = 2 + 8 * (_t - 1)
    NAMES.update({_b: f'conv{_t} last step', _b + 1: f'conv{_t} epilogue out', _b + 2: f'x{_t} publish: wait', _b + 3: f'x{_t} published',
                  _b + 4: f'x{_t} flags: wait', _b + 5: f'x{_t} tile issued', _b + 6: f'x{_t} first use: wait', _b + 7: f'x{_t} first use: go'})
NAMES.update({40: 'conv5 last step', 41: 'conv5 epilogue', 42: 'end'})


def run(n, h, w, nf=64, gc=32, wave4=0):
    lib = _lib.load()
    lib.sr_dev_fused_phase_clocks.argtypes = [C.c_void_p]
    lib.sr_dev_set_fused_wave4.argtypes = [C.c_int]
    lib.sr_dev_set_fused_wave4(wave4)
    print('four waves of four rows' if wave4 else 'eight waves of two rows')
    lib.sr_set_conv_chain(3)
    dev = torch.device('cuda')
    packs = _rdb(dev, nf, gc, 3)
    cat, nxt = _fresh(dev, n, nf, gc, h, w, 5)
    steps = _steps(cat, nxt, packs, nf, gc)
    dbg = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
    for it in range(3):
        lib.sr_dev_fused_phase_clocks(dbg.data_ptr() if it == 2 else None)
        H.conv3x3_chain_bf16(steps, None, 0)
    lib.sr_dev_fused_phase_clocks(None)
    torch.cuda.synchronize()
    t = dbg.cpu().view(256, 64).double()
    t = t[t[:, 42] > 0]
    rel = t - t[:, :1]
    print(f'n={n} {h}x{w}: {t.shape[0]} workgroups, last round; ticks of 10 ns since the round start (median / max over workgroups)')
    order = sorted(NAMES, key=lambda i: float(rel[:, i].median()))
    prev = 0.0
    for i in order:
        med, mx = float(rel[:, i].median()), float(rel[:, i].max())
        print(f'  {NAMES[i]:22s} {med:7.0f} {mx:7.0f}   (+{med - prev:6.0f})')
        prev = med
    xcd = t[:, 62].long() & 15
    for x in sorted(set(xcd.tolist())):
        sel = xcd == x
        st, en = t[sel, 60], t[sel, 61]
        ok = (st > 0) & (en > st)
        st, en = st[ok], en[ok]
        s0 = st.min()
        print(f'  XCD {x}: {int(ok.sum())} workgroups; starts +{float((st - s0).median()):.0f} (median) +{float((st - s0).max()):.0f} (last); ends '
              f'+{float((en - s0).min()):.0f} (first) +{float((en - s0).median()):.0f} (median) +{float((en - s0).max()):.0f} (last); own duration median {float((en - st).median()):.0f}')


if __name__ == '__main__':
    run(16, 128, 128)
    run(16, 128, 128, wave4=1)

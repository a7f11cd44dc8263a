"""Development aid: the fused dense block on eight waves of two rows vs four waves of four rows (sr_dev_set_fused_wave4): launch time
and shader-clock cycles per workgroup, forward block, batch 16 / 32 of 128x128 and four 544x544 tiler cells."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from image_restoration_amd import _lib, hip_ops as H
from test_chain_bf16_gpu import _fresh, _rdb, _steps

lib = _lib.load()
lib.sr_set_conv_chain(3)
lib.sr_dev_fused_phase_clocks.argtypes = [C.c_void_p]
lib.sr_dev_set_fused_wave4.argtypes = [C.c_int]
lib.sr_dev_set_chain_mids_scratch.argtypes = [C.c_int]
dev = torch.device('cuda')
for rnd in range(2):
    for n, h, w in ((16, 128, 128), (32, 128, 128), (4, 544, 544)):
        packs = _rdb(dev, 64, 32, 3)
        cat, nxt = _fresh(dev, n, 64, 32, h, w, 5)
        steps = _steps(cat, nxt, packs, 64, 32)
        row = []
        for w4, scratch in ((0, 0), (1, 0), (0, 1), (1, 1)):   # scratch: x1..x4 are not read afterwards (the inference forward): only their ring is stored
            lib.sr_dev_set_fused_wave4(w4)
            lib.sr_dev_set_chain_mids_scratch(scratch)
            for it in range(3):
                _, sync = H.conv3x3_chain_bf16(steps, None, 0)
            torch.cuda.synchronize()
            sync = torch.zeros_like(sync)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for it in range(40):
                H.conv3x3_chain_bf16(steps, sync, it)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 40
            dbg = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
            cyc = []
            for it in range(5):
                lib.sr_dev_fused_phase_clocks(dbg.data_ptr())
                H.conv3x3_chain_bf16(steps, sync, 100 + it)
                torch.cuda.synchronize()
                t = dbg.cpu().view(256, 64)
                cyc.append(float((t[:, 61] - t[:, 60]).double().median()))
            lib.sr_dev_fused_phase_clocks(None)
            row.append(f'{"four" if w4 else "eight"} waves{", ring-only stores" if scratch else ""} {us:.1f} us, {sorted(cyc)[2]:.0f} cycles, abort {int(sync[0])}')
        print(f'n={n} {h}x{w}: ' + '   '.join(row), flush=True)
lib.sr_dev_set_fused_wave4(0)
lib.sr_dev_set_chain_mids_scratch(0)

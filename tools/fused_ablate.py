"""Development aid: launch time of the fused dense block (batch 16 and 32 of 128x128, forward instance) for every ablated variant
library tools/fused_ablate.sh built (image_restoration_amd/lib/libsr_hip_abl<bits>.so).  One child process per library."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import torch
from image_restoration_amd import _lib, hip_ops as H
from test_chain_bf16_gpu import _fresh, _rdb, _steps
lib = _lib.load(); lib.sr_set_conv_chain(3)
lib.sr_dev_set_fused_wave4(int(os.environ.get('SR_ABL_WAVE4', '0')))
dev = torch.device('cuda')
out = []
for n in (16, 32):
    packs = _rdb(dev, 64, 32, 3)
    cat, nxt = _fresh(dev, n, 64, 32, 128, 128, 5)
    steps = _steps(cat, nxt, packs, 64, 32)
    for it in range(3):
        _, sync = H.conv3x3_chain_bf16(steps, None, 0)
    torch.cuda.synchronize()
    sync = torch.zeros_like(sync)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(40):
        H.conv3x3_chain_bf16(steps, sync, it)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 40
    # shader-clock cycles a workgroup spends in the launch (s_memtime stamps at its start and end; the stamps' own cost is in every variant)
    import ctypes as C
    lib.sr_dev_fused_phase_clocks.argtypes = [C.c_void_p]
    dbg = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
    cyc = []
    for it in range(6):
        lib.sr_dev_fused_phase_clocks(dbg.data_ptr())
        H.conv3x3_chain_bf16(steps, sync, 100 + it)
        torch.cuda.synchronize()
        t = dbg.cpu().view(256, 64)
        cyc.append(float((t[:, 61] - t[:, 60]).double().median()))
    lib.sr_dev_fused_phase_clocks(None)
    cyc = sorted(cyc)[len(cyc) // 2]
    out.append('%%d: %%.1f us, %%.0f cycles per workgroup' %% (n, us, cyc))
print('   '.join(out), ' abort', int(sync[0]))
''' % (ROOT, ROOT)

NAMES = {1: 'no step barriers', 2: 'no epilogues', 4: 'no LDS-DMA', 8: 'no operand reads', 16: 'no flag polling', 32: 'no epilogue stores', 64: 'no vmcnt waits', 128: 'LDS read bytes of four rows per wave', 256: 'no weight DMA', 512: 'no tile DMA', 1024: 'hand-offs through L2', 4096: 'no DMA of the x1..x4 tiles', 8192: 'two 16x16x32 MFMAs per 32x32x16'}
libs = sorted(glob.glob(os.path.join(ROOT, 'image_restoration_amd', 'lib', 'libsr_hip_abl*.so')), key=lambda p: int(re.findall(r'abl(\d+)', p)[0]))
for rnd in range(2):
    for w4 in os.environ.get('SR_ABL_LAYOUTS', '0,1').split(','):   # '1': a variant that only fits the four-wave instance's registers
        print('four waves of four rows' if w4 == '1' else 'eight waves of two rows')
        for p in libs:
            bits = int(re.findall(r'abl(\d+)', p)[0])
            what = ' + '.join(v for k, v in NAMES.items() if bits & k) or 'the product kernel'
            r = subprocess.run([sys.executable, '-c', CHILD], env=dict(os.environ, SR_HIP_LIB_PATH=p, SR_ABL_WAVE4=w4), capture_output=True, text=True,
                               timeout=300)
            print(f'abl {bits:2d} ({what}): {r.stdout.strip() or r.stderr.strip()[-300:]}', flush=True)
            if r.returncode != 0 or 'core dump' in (r.stdout + r.stderr):   # a faulting variant is not run a second time
                sys.exit(1)

import sys, os, time, ctypes as C
sys.path.insert(0, '/root/repo')
import torch
import image_restoration_amd as ira
from image_restoration_amd import _lib
from image_restoration_amd.utils import synth
cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32)
net = ira.build_network(dict(type='RRDBNet', **cfg)).cuda()
lib = _lib.load()
for dt in ('fp32', 'bf16'):
    net.set_compute_dtype(dt)
    c = net._cfg()
    st = torch.cuda.current_stream().cuda_stream
    for name, fn in (('fwd pack', lambda: (net._ensure_packed_bf16 if dt == 'bf16' else net._ensure_packed)(lib, c, st)),
                     ('dgrad pack', lambda: net._ensure_packed_dgrad(lib, c, st, dt == 'bf16'))):
        for _ in range(2):
            net.invalidate_packed(); fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            net.invalidate_packed(); fn()
        torch.cuda.synchronize()
        print(dt, name, f'{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms')

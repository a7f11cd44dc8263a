"""Times sr_rdb_wgrad_bf16 (all weight gradients of one dense block: wgrad_rdb_bf16_kernel + its two reduction launches).
usage: [XCD_MAP=0|1] python tools/wgrad_rdb_bench.py [n h w]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_restoration_amd import _lib

n, h, w = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (32, 128, 128)
lib = _lib.load()
nf, gc = 64, 32
cat = torch.randn(n, 12, h, w, 16, device='cuda').to(torch.bfloat16)
D = torch.randn(n, 12, h, w, 16, device='cuda').to(torch.bfloat16)
grads, ptrs = [], []
for k in range(1, 6):
    cout, cin = (nf if k == 5 else gc), nf + (k - 1) * gc
    grads += [torch.zeros(cout, cin, 3, 3, device='cuda'), torch.zeros(cout, device='cuda')]
ptrs = (C.c_void_p * 10)(*[g.data_ptr() for g in grads])
nbytes = lib.sr_rdb_wgrad_slab_bytes_bf16(n, h, w, nf, gc)
slab = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
flops = 2.0 * 9 * n * h * w * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64)
lib.sr_dev_set_wgrad_xcd_map.argtypes = [C.c_int]
lib.sr_dev_set_wgrad_xcd_map.restype = None
modes = (int(os.environ['XCD_MAP']),) if 'XCD_MAP' in os.environ else (0, 1, 0, 1, 0, 1)
for tri in modes:
    lib.sr_dev_set_wgrad_xcd_map(tri)
    def run():
        _lib.check(lib.sr_rdb_wgrad_bf16(cat.data_ptr(), D.data_ptr(), cat[0].numel(), n, h, w, nf, gc, ptrs, 0.2, 0, slab.data_ptr(), nbytes,
                                         torch.cuda.current_stream().cuda_stream), 'sr_rdb_wgrad_bf16')
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        run()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 50 * 1e3
    print(f'xcd_map {tri}: {us:.1f} us per block (kernel + reductions), {flops / us / 1e6:.0f} TFLOP/s   n={n} {h}x{w}', flush=True)

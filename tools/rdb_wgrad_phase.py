import sys, os, ctypes as C
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from image_restoration_amd import _lib, hip_ops as ops
lib = _lib.load()
def run_rdb(n, h, w, nf=64, gc=32):
    g = torch.Generator().manual_seed(0)
    ctot = nf + 4 * gc
    cat = ops.CB16(torch.randn(n, ctot // 16, h, w, 16, generator=g).to(torch.bfloat16).cuda())
    D = ops.CB16(torch.randn(n, ctot // 16, h, w, 16, generator=g).to(torch.bfloat16).cuda())
    grads = []
    for k in range(1, 6):
        cout, cin = (nf if k == 5 else gc), nf + (k - 1) * gc
        grads += [torch.zeros((cout, cin, 3, 3), device='cuda'), torch.zeros((cout,), device='cuda')]
    arr = (C.c_void_p * 10)(*[t.data_ptr() for t in grads])
    nbytes = lib.sr_rdb_wgrad_slab_bytes_bf16(n, h, w, nf, gc)
    slab = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    dbg = torch.zeros(1 << 18, dtype=torch.int64, device='cuda')
    lib.sr_dev_wgrad_bf16_phase_clocks.argtypes = [C.c_void_p]
    for it in range(3):
        lib.sr_dev_wgrad_bf16_phase_clocks(dbg.data_ptr() if it == 2 else None)
        _lib.check(lib.sr_rdb_wgrad_bf16(cat.ptr, D.ptr, cat.img_stride, n, h, w, nf, gc, arr, 0.04, 0, slab.data_ptr(), nbytes, None), 'rdb')
    lib.sr_dev_wgrad_bf16_phase_clocks(None)
    torch.cuda.synchronize()
    t = dbg.cpu().view(-1, 8)
    slot = (torch.arange(t.shape[0]) % 8)[t[:, 0] > 1e9]
    t = t[t[:, 0] > 1e9].double()
    t0 = t[:, 0].min()
    for item in range(4):
        u = t[t[:, 7] == item]
        if len(u) == 0: continue
        sl = slot[t[:, 7] == item]
        print('   wait + barrier per wave slot (cycles, of total): ' + ' '.join(f'{float(u[sl == k][:, 2].mean()):.0f}' for k in range(8)), flush=True)
        print(f'rdb item {item}: waves={len(u)} start spread {float(u[:,0].max()-t0):.0f} | first wait {float(u[:,1].mean()):.0f} | later waits '
              f'{float(u[:,2].mean()):.0f} (min {float(u[:,2].min()):.0f} max {float(u[:,2].max()):.0f}) | issue {float(u[:,6].mean()):.0f} | loop end {float(u[:,3].mean()):.0f} | total {float(u[:,4].mean()):.0f} '
              f'| steps {float(u[:,5].mean()):.0f} | end max {float((u[:,0]+u[:,4]).max()-t0):.0f}', flush=True)
run_rdb(32, 128, 128)

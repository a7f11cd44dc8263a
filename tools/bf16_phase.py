"""Development aid: per-workgroup phase clocks of conv_bf16_kernel (prologue / compute / barrier wait / epilogue)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import torch
from image_restoration_amd import hip_ops as ops, _lib

def run(n, cin, cout, h, w, res=False):
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    src = ops.CB16(torch.randn(n, cin // 16, h, w, 16, generator=g).to(torch.bfloat16).cuda())
    pc = ops.PackedConvBF16((torch.randn(cout, cin, 3, 3, generator=g) * 0.02).cuda(), torch.zeros(cout).cuda())
    out = ops.CB16.zeros(n, cout, h, w, 'cuda')
    r1 = ops.CB16.zeros(n, cout, h, w, 'cuda') if res else None
    dbg = torch.zeros(1 << 20, dtype=torch.int64, device='cuda')
    d = _lib.ConvDesc()
    d.in_, d.in_img_stride, d.cin_pad, d.in_h, d.in_w = src.ptr, src.img_stride, cin, h, w
    d.wpacked, d.bpacked, d.cout = pc.w.data_ptr(), pc.b.data_ptr(), cout
    d.out, d.out_img_stride, d.n, d.act_slope, d.alpha = out.ptr, out.img_stride, n, 0.2, 1.0
    if res:
        d.res1, d.res1_img_stride, d.beta1 = r1.ptr, r1.img_stride, 0.2
    lib.sr_dev_conv_bf16_phase_clocks.argtypes = [C.c_void_p]
    for it in range(3):
        lib.sr_dev_conv_bf16_phase_clocks(dbg.data_ptr() if it == 2 else None)
        _lib.check(lib.sr_conv3x3_bf16(C.byref(d), None), 'conv')
    lib.sr_dev_conv_bf16_phase_clocks(None)
    torch.cuda.synchronize()
    t = dbg.cpu().view(-1, 8)
    t = t[t[:, 0] > 1e9].double()
    t0 = t[:, 0].min()
    print(f'n={n} cin={cin} cout={cout} res={res}: waves={len(t)} start spread {float(t[:,0].max()-t0):.0f} '
          f'| prologue {float(t[:,1].mean()):.0f} | compute {float(t[:,2].mean()):.0f} | barrier-wait {float(t[:,3].mean()):.0f} '
          f'| loop total {float(t[:,4].mean()):.0f} | epilogue {float(t[:,5].mean()):.0f} | end max {float((t[:,0]+t[:,4]+t[:,5]).max()-t0):.0f} (shader cycles)')


def run_wgrad(n, cin, cout, h, w, first_seg=None, seg=0):
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    cin_pad = lib.sr_conv3x3_cin_pad16(cin, cin if first_seg is None else first_seg, seg)
    src = ops.CB16(torch.randn(n, cin_pad // 16, h, w, 16, generator=g).to(torch.bfloat16).cuda())
    dy = ops.CB16(torch.randn(n, (cout + 15) // 16, h, w, 16, generator=g).to(torch.bfloat16).cuda())
    dbg = torch.zeros(1 << 18, dtype=torch.int64, device='cuda')
    lib.sr_dev_wgrad_bf16_phase_clocks.argtypes = [C.c_void_p]
    for it in range(3):
        lib.sr_dev_wgrad_bf16_phase_clocks(dbg.data_ptr() if it == 2 else None)
        ops.conv3x3_wgrad_bf16(src, dy, cout, cin, first_seg, seg)
    lib.sr_dev_wgrad_bf16_phase_clocks(None)
    torch.cuda.synchronize()
    t = dbg.cpu().view(-1, 8)
    t = t[t[:, 0] > 1e9].double()
    t0 = t[:, 0].min()
    print(f'wgrad n={n} cin={cin} cout={cout}: waves={len(t)} (last launch group) start spread {float(t[:,0].max()-t0):.0f} | '
          f'first wait {float(t[:,1].mean()):.0f} | later waits {float(t[:,2].mean()):.0f} | loop end {float(t[:,3].mean()):.0f} '
          f'| total {float(t[:,4].mean()):.0f} | steps {float(t[:,5].mean()):.0f} | end max {float((t[:,0]+t[:,4]).max()-t0):.0f} cycles')


def run_rdb(n, h, w, nf=64, gc=32):
    lib = _lib.load()
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
    t = t[t[:, 0] > 1e9].double()
    t0 = t[:, 0].min()
    for item in range(4):
        u = t[t[:, 7] == item]
        if len(u) == 0: continue
        print(f'rdb item {item}: waves={len(u)} start spread {float(u[:,0].max()-t0):.0f} | first wait {float(u[:,1].mean()):.0f} | later waits '
              f'{float(u[:,2].mean()):.0f} | issue {float(u[:,6].mean()):.0f} | loop end {float(u[:,3].mean()):.0f} | total {float(u[:,4].mean()):.0f} '
              f'| steps {float(u[:,5].mean()):.0f} | end max {float((u[:,0]+u[:,4]).max()-t0):.0f}')

run_rdb(16, 128, 128)
run_wgrad(16, 64, 32, 128, 128)
run_wgrad(16, 128, 32, 128, 128, 64, 32)
run_wgrad(16, 160, 32, 128, 128, 64, 32)
run_wgrad(16, 128, 64, 128, 128, 64, 32)
run(16, 160, 32, 128, 128)
run(16, 192, 64, 128, 128, True)
run(16, 64, 64, 128, 128)

"""Development aid: one residual dense block in bf16, conv by conv vs the persistent chain launch (sr_conv3x3_chain_bf16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch
from image_restoration_amd import hip_ops as H
from test_chain_bf16_gpu import _fresh, _rdb, _steps


def bench(n, h, w, iters=30, nf=64, gc=32):
    dev = torch.device('cuda')
    packs = _rdb(dev, nf, gc, 3)
    cat, nxt = _fresh(dev, n, nf, gc, h, w, 5)
    steps = _steps(cat, nxt, packs, nf, gc)
    res = {}
    for name in ('conv-by-conv', 'chain'):
        sync = None
        for it in range(3):
            if name == 'chain':
                _, sync = H.conv3x3_chain_bf16(steps, None, 0)
            else:
                for src, pc, out, kw in steps:
                    H.conv3x3_bf16(src, pc, out, **kw)
        torch.cuda.synchronize()
        sync = torch.zeros_like(sync) if sync is not None else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(iters):
            if name == 'chain':
                H.conv3x3_chain_bf16(steps, sync, it)
            else:
                for src, pc, out, kw in steps:
                    H.conv3x3_bf16(src, pc, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / iters
        assert sync is None or int(sync[0]) == 0
    fl = 2.0 * 9 * n * h * w * sum((nf + k * gc) * (nf if k == 4 else gc) for k in range(5))
    print(f'n={n} {h}x{w}: conv-by-conv {res["conv-by-conv"]:.1f} us ({fl / res["conv-by-conv"] / 1e6:.0f} TF)   chain {res["chain"]:.1f} us '
          f'({fl / res["chain"] / 1e6:.0f} TF)   x{res["conv-by-conv"] / res["chain"]:.2f}', flush=True)


def bench_f32(n, h, w, iters=10, nf=64, gc=32):
    import test_chain_f32_gpu as T
    dev = torch.device('cuda')
    packs = T._rdb(dev, nf, gc, 3)
    cat, nxt = T._fresh(dev, n, nf, gc, h, w, 5)
    steps = T._steps(cat, nxt, packs, nf, gc)
    res = {}
    for name in ('conv-by-conv', 'chain'):
        sync = None
        for it in range(2):
            if name == 'chain':
                _, sync = H.conv3x3_chain(steps, None, 0)
            else:
                for src, pc, out, kw in steps:
                    H.conv3x3(src, pc, out, **kw)
        torch.cuda.synchronize()
        sync = torch.zeros_like(sync) if sync is not None else None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(iters):
            if name == 'chain':
                H.conv3x3_chain(steps, sync, it)
            else:
                for src, pc, out, kw in steps:
                    H.conv3x3(src, pc, out, **kw)
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / iters
        assert sync is None or int(sync[0]) == 0
    fl = 2.0 * 9 * n * h * w * sum((nf + k * gc) * (nf if k == 4 else gc) for k in range(5))
    print(f'fp32 n={n} {h}x{w}: conv-by-conv {res["conv-by-conv"]:.1f} us ({fl / res["conv-by-conv"] / 1e6:.1f} TF)   chain {res["chain"]:.1f} us '
          f'({fl / res["chain"] / 1e6:.1f} TF)   x{res["conv-by-conv"] / res["chain"]:.3f}', flush=True)


if __name__ == '__main__':
    from image_restoration_amd import _lib
    mode = int(os.environ.get('CHAIN_MODE', '1'))
    _lib.check(_lib.load().sr_set_conv_chain(mode), 'chain')
    _lib.check(_lib.load().sr_set_conv_chain_f32(1), 'chain')
    print('bf16 chain mode', mode)
    if os.environ.get('CHAIN_F32'):
        bench_f32(16, 128, 128)
        bench_f32(32, 128, 128)
    bench(16, 128, 128)
    bench(4, 128, 128)
    bench(32, 128, 128)
    bench(4, 544, 544, iters=10)
    bench(1, 544, 544, iters=10)

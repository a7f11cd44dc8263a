"""bf16 / fp32 whole-network forward vs sr_set_forward_groups (image groups on concurrent streams)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_restoration_amd as ira
from image_restoration_amd import _lib

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = torch.device('cuda:0')
net = ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32,
                             compute_dtype=dtype)).to(dev).eval()
lib = _lib.load()
for shape in ((16, 128, 128), (4, 544, 544), (32, 128, 128)):
    x = torch.rand(shape[0], 3, shape[1], shape[2], device=dev)
    ref = None
    for g in (1, 2, 3, 4):
        _lib.check(lib.sr_set_forward_groups(g), 'groups')
        with torch.no_grad():
            for _ in range(2):
                y = net(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8):
                y = net(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 8
        if ref is None:
            ref = y.clone()
        print(f'{dtype} {shape} groups={g}: {dt * 1e3:.2f} ms  {shape[0] / dt:.1f} img/s  same={bool(torch.equal(y, ref))}', flush=True)

"""Single-image latency of the generator (the reference's serving case: one small plate crop per call).
usage: python tools/latency.py [h w] [dtype]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_restoration_amd as ira
h = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dt = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
net = ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32,
                             compute_dtype=dt)).cuda().eval()
for n in (1, 4):
    x = torch.rand(n, 3, h, w, device='cuda')
    with torch.no_grad():
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K = 30
        for _ in range(K):
            y = net(x)
        torch.cuda.synchronize()
    print(f'{dt} batch {n} of {h}x{w}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms per call')

"""Experiment: does splitting the batch over S concurrent HIP streams hide the per-launch ramp / tail of the bf16 convs?"""
import sys
import time

import torch

sys.path.insert(0, '.')
import image_restoration_amd as ira  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
dev = torch.device('cuda:0')
cfg = dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32, compute_dtype=dtype)
for S in (1, 2, 4):
    nets = [ira.build_network(dict(cfg)).to(dev).eval() for _ in range(S)]
    xs = [torch.rand(16 // S, 3, 128, 128, device=dev) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]

    def step():
        for net, x, st in zip(nets, xs, streams):
            with torch.cuda.stream(st), torch.no_grad():
                net(x)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    print(f'{dtype} streams={S}: {dt * 1e3:.2f} ms per 16 tiles = {16 / dt:.1f} img/s', flush=True)

"""Development aid: inference throughput of the 23-block network against the batch size (is the working set cache-resident?)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_restoration_amd as ira
from image_restoration_amd import _lib
from image_restoration_amd.utils import synth

CFG = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32)
dev = torch.device('cuda')
net = ira.build_network(dict(type='RRDBNet', **CFG)).to(dev).eval()
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(0, **CFG).items()}, strict=True)
for dtype in sys.argv[1:] or ['bf16', 'fp32']:
    net.set_compute_dtype(dtype)
    for groups in ((0, 1) if dtype == 'bf16' else (0,)):
        _lib.check(_lib.load().sr_set_forward_groups(groups), 'groups')
        for b in (2, 4, 6, 8, 12, 16, 24, 32):
            x = torch.from_numpy(synth.uniform_input(1, (b, 3, 128, 128))).to(dev)
            with torch.no_grad():
                for _ in range(3):
                    net(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                n = 10
                for _ in range(n):
                    net(x)
                torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f'{dtype} groups={groups} batch {b:3d}: {dt * 1e3:8.3f} ms  {b / dt:8.1f} img/s', flush=True)

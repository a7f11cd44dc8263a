"""Development aid: which host call sites launch torch fill / copy kernels in a bf16 UNet discriminator step."""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_restoration_amd as ira
from torch.utils._python_dispatch import TorchDispatchMode

dev = torch.device('cuda:0')
net = ira.build_network(dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=64, skip_connection=True, compute_dtype='bf16')).to(dev)
x = torch.rand(4, 3, 256, 256, device=dev, requires_grad=True)
net(x).mean().backward()
sites = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(s in name for s in ('fill', 'zero', 'copy', 'clone', 'add', 'contiguous', 'mul')):
            big = [tuple(a.shape) for a in args if torch.is_tensor(a) and a.numel() > 1 << 16]
            if big:
                st = [f'{os.path.basename(f.filename)}:{f.lineno}' for f in traceback.extract_stack()[:-1]
                      if 'image_restoration_amd' in f.filename or 'tools' in f.filename]
                sites[(name, st[-1] if st else '<autograd engine>', str(big[0]))] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    net(x).mean().backward()
for k, v in sites.most_common():
    print(v, k)

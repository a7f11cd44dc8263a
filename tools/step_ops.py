"""Development aid: torch (aten) ops issued by one ESRGAN step outside libsr_hip.so, by host call site.
usage: python tools/step_ops.py <yml> [batch] [lq]"""
import sys, os, collections, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from image_restoration_amd.utils.options import parse
from image_restoration_amd.models import build_model
from image_restoration_amd.utils import synth

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
opt = parse(os.path.join(root, sys.argv[1]), root, is_train=True)
opt.update(dist=False, rank=0, world_size=1, num_gpu=1)
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
LQ = int(sys.argv[3]) if len(sys.argv) > 3 else 32
model = build_model(opt)
lq = torch.from_numpy(synth.uniform_input(1, (B, 3, LQ, LQ)))
gt = torch.from_numpy(synth.uniform_input(2, (B, 3, 4 * LQ, 4 * LQ)))


def step(i):
    model.update_learning_rate(i, warmup_iter=-1)
    model.feed_data({'lq': lq, 'gt': gt})
    model.optimize_parameters(i)


for i in (1, 2):
    step(i)
sites = collections.Counter()
SKIP = ('aten.view', 'aten.detach', 'aten._unsafe_view', 'aten.t.', 'aten.alias', 'aten.empty', 'aten.as_strided', 'aten.slice',
        'aten.select', 'aten.expand', 'aten.reshape', 'aten.unsqueeze', 'aten.squeeze', 'aten.transpose', 'aten.permute')


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            st = [f'{os.path.basename(f.filename)}:{f.lineno}' for f in traceback.extract_stack()[:-1]
                  if 'image_restoration_amd' in f.filename]
            sites[(name, st[-1] if st else '<autograd engine>')] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step(3)
for k, v in sites.most_common(40):
    print(v, k)

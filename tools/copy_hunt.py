"""Which host calls of a training step end in a device copy / fill?  torch.profiler over a few steps of the recipe; prints the
operator table (CPU side) and the copy / memset rows.  usage: python tools/copy_hunt.py [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from image_restoration_amd.utils.options import parse
from image_restoration_amd.models import build_model
from image_restoration_amd.utils import synth

dtype = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
c3 = len(sys.argv) > 2 and sys.argv[2] == 'c3'   # the full-size step: U-Net discriminator, 128x128 LR patches
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
opt = parse(os.path.join(root, 'training_config/train_rrdbnet_esrgan_x4_mi355x.yml'), root, is_train=True)
opt.update(dist=False, rank=0, world_size=1, num_gpu=1)
opt['network_g']['compute_dtype'] = dtype
if c3:
    opt['network_d'] = dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=64, skip_connection=True, compute_dtype=dtype)
else:
    opt['network_d']['compute_dtype'] = dtype
model = build_model(opt)
LQ = 128 if c3 else 32
lq = torch.from_numpy(synth.uniform_input(1, (32, 3, LQ, LQ))).cuda()
gt = torch.from_numpy(synth.uniform_input(2, (32, 3, 4 * LQ, 4 * LQ))).cuda()


def step(i):
    model.update_learning_rate(i, warmup_iter=-1)
    model.feed_data({'lq': lq, 'gt': gt})
    model.optimize_parameters(i)


for i in range(1, 4):
    step(i)
torch.cuda.synchronize()
K = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    for i in range(4, 4 + K):
        step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='self_cpu_time_total', row_limit=70, max_name_column_width=60))

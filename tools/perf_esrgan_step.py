"""Times full ESRGANModel.optimize_parameters steps (G fwd/bwd, D, losses, both Adam steps, EMA) on synthetic batches.
usage: python tools/perf_esrgan_step.py <yml> [batch] [lq_size] [iters] [compute_dtype] [disc: vgg|unet] [disc compute_dtype] [perceptual compute_dtype]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_restoration_amd.utils.options import parse
from image_restoration_amd.models import build_model
from image_restoration_amd.utils import synth

yml = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
LQ = int(sys.argv[3]) if len(sys.argv) > 3 else 32
K = int(sys.argv[4]) if len(sys.argv) > 4 else 5
DT = sys.argv[5] if len(sys.argv) > 5 else None
DISC = sys.argv[6] if len(sys.argv) > 6 else None
DDT = sys.argv[7] if len(sys.argv) > 7 else None
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
opt = parse(os.path.join(root, yml), root, is_train=True)
opt['dist'] = False
opt['rank'], opt['world_size'] = 0, 1
opt['num_gpu'] = 1
if DT:
    opt['network_g']['compute_dtype'] = DT
if DISC == 'unet':
    opt['network_d'] = dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=64, skip_connection=True)
if DDT:
    opt['network_d']['compute_dtype'] = DDT
if len(sys.argv) > 8 and opt['train'].get('perceptual_opt'):
    opt['train']['perceptual_opt']['compute_dtype'] = sys.argv[8]
model = build_model(opt)
lq = torch.from_numpy(synth.uniform_input(1, (B, 3, LQ, LQ)))
gt = torch.from_numpy(synth.uniform_input(2, (B, 3, 4 * LQ, 4 * LQ)))
def step(i):
    model.update_learning_rate(i, warmup_iter=-1)
    model.feed_data({'lq': lq, 'gt': gt})
    model.optimize_parameters(i)
for i in range(1, 3):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(3, 3 + K):
    step(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
log = model.get_current_log()
print(f'{os.path.basename(yml)} dtype={opt["network_g"].get("compute_dtype", "fp32")} disc={opt["network_d"]["type"]} batch {B} lq {LQ}: '
      f'{dt * 1e3:.1f} ms/step = {B / dt:.1f} img/s   losses: ' + ' '.join(f'{k}={v:.4g}' for k, v in log.items()))

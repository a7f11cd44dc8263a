"""Is a training step bound by the host that issues it or by the GPU?  Times K steps three ways: wall (synchronised at both ends),
host issue time (the python thread alone, no synchronisation inside), and the GPU's own time between the first and the last kernel
(events).  usage: python tools/issue_time.py <yml> [batch] [lq] [iters] [dtype] [disc dtype]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_restoration_amd.utils.options import parse
from image_restoration_amd.models import build_model
from image_restoration_amd.utils import synth

yml = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
LQ = int(sys.argv[3]) if len(sys.argv) > 3 else 32
K = int(sys.argv[4]) if len(sys.argv) > 4 else 10
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
opt = parse(os.path.join(root, yml), root, is_train=True)
opt.update(dist=False, rank=0, world_size=1, num_gpu=1)
if len(sys.argv) > 5:
    opt['network_g']['compute_dtype'] = sys.argv[5]
if len(sys.argv) > 6:
    opt['network_d']['compute_dtype'] = sys.argv[6]
if os.environ.get('CHAIN_F32'):   # development: the fp32 dense block as one persistent launch (off by default)
    from image_restoration_amd import _lib
    _lib.check(_lib.load().sr_set_conv_chain_f32(int(os.environ['CHAIN_F32'])), 'sr_set_conv_chain_f32')
if os.environ.get('ROWS8'):   # development: 0 = the fused dense block on 16-row tiles only
    import ctypes as _C
    from image_restoration_amd import _lib as _l
    _l.load().sr_dev_set_fused_rows8.argtypes = [_C.c_int]
    _l.load().sr_dev_set_fused_rows8(int(os.environ['ROWS8']))
if os.environ.get('WGRAD_TARGET'):   # development: workgroups per fp32 weight-gradient launch the row split aims at
    from image_restoration_amd import _lib as _l2
    _l2.load().sr_dev_set_wgrad_f32_target(int(os.environ['WGRAD_TARGET']))
if os.environ.get('RDB_WGRAD'):   # development: "0" = fp32 dense-block weight gradients one tile-group set per launch; "1,<wgs>" = target workgroups
    import ctypes as _C3
    from image_restoration_amd import _lib as _l3
    _v = os.environ['RDB_WGRAD'].split(',')
    _l3.load().sr_dev_set_rdb_wgrad_f32.argtypes = [_C3.c_int, _C3.c_int]
    _l3.load().sr_dev_set_rdb_wgrad_f32(int(_v[0]), int(_v[1]) if len(_v) > 1 else 0)
if os.environ.get('FWD_GROUPS'):   # development: "<groups>,<min workgroups per group launch>" for the fp32 forward
    from image_restoration_amd import _lib as _l4
    _g = os.environ['FWD_GROUPS'].split(',')
    _l4.check(_l4.load().sr_set_forward_groups(int(_g[0])), 'sr_set_forward_groups')
    if len(_g) > 1:
        _l4.load().sr_dev_set_group_min_wgs(int(_g[1]))
if os.environ.get('VGG_LANE'):   # development: 0 = the VGG discriminator's weight gradients on the caller's stream
    from image_restoration_amd import _lib as _l6
    _l6.load().sr_dev_set_vgg_lane(int(os.environ['VGG_LANE']))
model = build_model(opt)
lq = torch.from_numpy(synth.uniform_input(1, (B, 3, LQ, LQ))).cuda()
gt = torch.from_numpy(synth.uniform_input(2, (B, 3, 4 * LQ, 4 * LQ))).cuda()


def step(i):
    model.update_learning_rate(i, warmup_iter=-1)
    model.feed_data({'lq': lq, 'gt': gt})
    model.optimize_parameters(i)


for i in range(1, 4):
    step(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
e0.record()
for i in range(4, 4 + K):
    step(i)
e1.record()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_wall = time.perf_counter() - t0
print(f'{os.path.basename(yml)} batch {B} lq {LQ}: wall {t_wall / K * 1e3:.2f} ms/step, host issue {t_issue / K * 1e3:.2f} ms/step, '
      f'GPU first-to-last {e0.elapsed_time(e1) / K:.2f} ms/step')
if os.environ.get('CPROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    for i in range(4 + K, 4 + 2 * K):
        step(i)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)

"""Times generator forward+backward (argv: batch [fp32|bf16] [forward groups, 0 = default]) and prints per-kernel aggregates from the launch profiler."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import image_restoration_amd as ira
from image_restoration_amd import _lib
from image_restoration_amd.utils import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
DT = sys.argv[2] if len(sys.argv) > 2 else 'fp32'
G = int(sys.argv[3]) if len(sys.argv) > 3 else 0
_lib.check(_lib.load().sr_set_forward_groups(G), 'sr_set_forward_groups')
if os.environ.get('RDB_WGRAD'):   # development: "0" = fp32 dense-block weight gradients one tile-group set per launch; "1,<wgs>" = target workgroups
    _v = os.environ['RDB_WGRAD'].split(',')
    _lib.load().sr_dev_set_rdb_wgrad_f32.argtypes = [C.c_int, C.c_int]
    _lib.load().sr_dev_set_rdb_wgrad_f32(int(_v[0]), int(_v[1]) if len(_v) > 1 else 0)
cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32)
dev = torch.device('cuda')
net = ira.build_network(dict(type='RRDBNet', **cfg)).to(dev)
net.set_compute_dtype(DT)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(0, **cfg).items()})
x = torch.from_numpy(synth.uniform_input(1, (B, 3, 128, 128))).to(dev)
gy = torch.from_numpy(synth.signed_input(2, (B, 3, 512, 512))).to(dev)
def step():
    for p in net.parameters(): p.grad = None
    y = net(x)
    y.backward(gy)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); K = 6
for _ in range(K): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f'{DT} batch {B} groups {G}: fwd+bwd {dt*1e3:.1f} ms  -> {B/dt:.1f} img/s ; ideal(3x fwd flops @157.3TF) {3*5.8743e11*B/157.3e12*1e3:.1f} ms')
lib = _lib.load(); cap = 8192; recs = (_lib.LaunchRecord * cap)(); n = C.c_int(0)
lib.sr_profile_start(cap); step(); lib.sr_profile_stop(recs, cap, C.byref(n))
agg = {}
for r in recs[:n.value]:
    a = agg.setdefault(r.kernel_id, [0, 0.0, 0.0]); a[0] += 1; a[1] += r.ms; a[2] += r.flops
tot = sum(a[1] for a in agg.values())
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f'  id {k:2d} {lib.sr_kernel_name(k).decode():44s} launches {a[0]:4d} total {a[1]:8.2f} ms  {a[2]/a[1]/1e9:7.1f} TFLOP/s')
print('  sum of profiled launches', round(tot, 1), 'ms')

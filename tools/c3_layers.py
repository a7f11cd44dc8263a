"""Development aid: the conv / weight-gradient launches of one C3 training step (bf16 RRDBNet + bf16 UNetDiscriminatorSN, batch 32 of 128x128),
grouped by (kernel, cin, cout, h, w): time, TFLOP/s and algorithmic GB/s per layer shape (HIP events around every launch: sr_profile_*).
usage: python tools/c3_layers.py [recipe]     (recipe: the reference's own step — batch 32 of 32x32 patches, VGG discriminator, all bf16)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from image_restoration_amd import _lib


def main():
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    from image_restoration_amd.models import build_model
    from image_restoration_amd.utils import synth
    from image_restoration_amd.utils.options import parse, set_random_seed
    recipe = len(sys.argv) > 1 and sys.argv[1] == 'recipe'
    yml = 'train_rrdbnet_esrgan_x4_mi355x.yml' if recipe else 'train_rrdbnet_esrgan_x4_mi355x_bf16_unet.yml'
    opt = parse(os.path.join(bench.ROOT, 'training_config', yml), bench.ROOT, is_train=True)
    opt['dist'], opt['rank'], opt['world_size'], opt['num_gpu'] = False, 0, 1, 1
    opt['network_g']['compute_dtype'] = 'bf16'
    if recipe:
        opt['network_d']['compute_dtype'] = 'bf16'
    else:
        opt['network_d'] = dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=64, skip_connection=True, compute_dtype='bf16')
    set_random_seed(opt['manual_seed'])
    model = build_model(opt)
    b, lq = (32, 32) if recipe else (32, 128)
    data = {'lq': torch.from_numpy(synth.uniform_input(100, (b, 3, lq, lq))).to(dev),
            'gt': torch.from_numpy(synth.uniform_input(200, (b, 3, 4 * lq, 4 * lq))).to(dev)}
    for it in range(2):
        model.update_learning_rate(it + 1, warmup_iter=-1)
        model.feed_data(data)
        model.optimize_parameters(it + 1)
    torch.cuda.synchronize()
    lib = _lib.load()
    cap = 32768
    recs = (_lib.LaunchRecord * cap)()
    n = C.c_int(0)
    _lib.check(lib.sr_profile_start(cap), 'start')
    model.feed_data(data)
    model.optimize_parameters(3)
    torch.cuda.synchronize()
    _lib.check(lib.sr_profile_stop(recs, cap, C.byref(n)), 'stop')
    agg = {}
    for r in recs[:n.value]:
        k = (lib.sr_kernel_name(r.kernel_id).decode(), r.cin, r.cout, r.n, r.h, r.w)
        a = agg.setdefault(k, [0, 0.0, 0.0, 0.0])
        a[0] += 1; a[1] += r.ms; a[2] += r.flops; a[3] += r.bytes
    tot = sum(a[1] for a in agg.values())
    print(f'{n.value} launches, {tot:.1f} ms of profiled launches (serialised by the events)')
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f'{a[1]:8.2f} ms {a[0]:4d}x {a[1] / a[0] * 1e3:8.1f} us  {a[2] / a[1] / 1e9:7.0f} TF/s {a[3] / a[1] / 1e6:7.0f} GB/s  {k}')


if __name__ == '__main__':
    main()

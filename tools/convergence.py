"""End-to-end learning check on the HIP path: SRModel (L1) on smooth synthetic images, LQ = 4x4 box-filtered GT.
usage: python tools/convergence.py [fp32|bf16] [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from image_restoration_amd.utils.synth import smooth_pairs as smooth_batch
from image_restoration_amd.models import build_model
from image_restoration_amd.metrics import psnr_device


def run(dtype, iters):
    torch.manual_seed(0)
    opt = dict(name='conv', model_type='SRModel', scale=4, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
               network_g=dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=4, num_feat=32, num_block=2, num_grow_ch=16,
                              compute_dtype=dtype),
               path=dict(pretrain_network_g=None, strict_load_g=True),
               train=dict(ema_decay=0, optim_g=dict(type='Adam', lr=1e-3, weight_decay=0, betas=[0.9, 0.99]),
                          scheduler=dict(type='MultiStepLR', milestones=[10 ** 6], gamma=0.5), total_iter=iters, warmup_iter=-1,
                          pixel_opt=dict(type='L1Loss', loss_weight=1.0, reduction='mean')))
    model = build_model(opt)
    vlq, vgt = smooth_batch(999, 4, 96)
    hist = []
    for it in range(1, iters + 1):
        lq, gt = smooth_batch(it, 8, 96)
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_data({'lq': lq, 'gt': gt})
        model.optimize_parameters(it)
        if it == 1 or it % 50 == 0:
            model.feed_data({'lq': vlq, 'gt': vgt})
            model.test()
            ps = sum(psnr_device(model.output, model.gt, 4)) / 4
            hist.append((it, float(model.get_current_log()['l_pix']), ps))
    base = sum(psnr_device(F.interpolate(vlq, scale_factor=4, mode='nearest').cuda(), vgt.cuda(), 4)) / 4
    return hist, base


if __name__ == '__main__':
    dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    hist, base = run(dtype, iters)
    for it, l, p in hist:
        print(f'{dtype} iter {it:4d}  l_pix {l:.4f}  val PSNR {p:.2f} dB')
    print(f'nearest-upsampled LQ: {base:.2f} dB')

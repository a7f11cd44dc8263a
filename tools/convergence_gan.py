"""End-to-end check of the GAN recipe on the HIP path: ESRGANModel (L1 + perceptual + relativistic GAN; VGG discriminator, or the
spectrally normalised U-Net of the C3 step) on smooth synthetic pairs, fp32 vs bf16 for every component.
usage: python tools/convergence_gan.py [fp32|bf16] [iters] [vgg|unet]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from image_restoration_amd.metrics import psnr_device
from image_restoration_amd.models import build_model
from image_restoration_amd.utils.synth import smooth_pairs


def run(dtype, iters, disc='vgg'):
    torch.manual_seed(0)
    adam = dict(type='Adam', lr=5e-4, weight_decay=0, betas=[0.9, 0.99])
    opt = dict(name='gan', model_type='ESRGANModel', scale=4, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
               network_g=dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=4, num_feat=32, num_block=2, num_grow_ch=16, compute_dtype=dtype),
               network_d=(dict(type='VGGStyleDiscriminator128', num_in_ch=3, num_feat=16, compute_dtype=dtype) if disc == 'vgg' else
                          dict(type='UNetDiscriminatorSN', num_in_ch=3, num_feat=16, skip_connection=True, compute_dtype=dtype)),
               path=dict(pretrain_network_g=None, strict_load_g=True, pretrain_network_d=None),
               train=dict(ema_decay=0, optim_g=dict(adam), optim_d=dict(adam),
                          scheduler=dict(type='MultiStepLR', milestones=[10 ** 6], gamma=0.5), total_iter=iters, warmup_iter=-1,
                          pixel_opt=dict(type='L1Loss', loss_weight=1.0, reduction='mean'),
                          perceptual_opt=dict(type='PerceptualLoss', allow_random_init=True, layer_weights={'conv3_4': 1.0}, vgg_type='vgg19', perceptual_weight=0.05,
                                              style_weight=0, criterion='l1', compute_dtype=dtype),
                          gan_opt=dict(type='GANLoss', gan_type='vanilla', real_label_val=1.0, fake_label_val=0.0, loss_weight=5e-3),
                          net_d_iters=1, net_d_init_iters=0))
    model = build_model(opt)
    vlq, vgt = smooth_pairs(999, 4, 128)
    hist = []
    for it in range(1, iters + 1):
        lq, gt = smooth_pairs(it, 8, 128)
        model.update_learning_rate(it, warmup_iter=-1)
        model.feed_data({'lq': lq, 'gt': gt})
        model.optimize_parameters(it)
        if it == 1 or it % 50 == 0:
            log = model.get_current_log()
            model.feed_data({'lq': vlq, 'gt': vgt})
            model.test()
            hist.append((it, sum(psnr_device(model.output, model.gt, 4)) / 4, {k: round(float(v), 4) for k, v in log.items()}))
    return hist


if __name__ == '__main__':
    dtype = sys.argv[1] if len(sys.argv) > 1 else 'fp32'
    disc = sys.argv[3] if len(sys.argv) > 3 else 'vgg'
    for it, p, log in run(dtype, int(sys.argv[2]) if len(sys.argv) > 2 else 250, disc):
        print(f'{dtype} {disc} iter {it:4d} val PSNR {p:6.2f} dB  {log}')

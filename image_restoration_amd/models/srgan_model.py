"""SRGANModel: generator + discriminator with the plain (non-relativistic) GAN loss.

Counterpart of basicsr/models/srgan_model.py:15-143: same option keys, requires_grad toggling on D (:81-83,
:111-113), step order (G: pix + gan -> backward -> Adam; D: real backward, fake backward -> Adam), log keys."""
from collections import OrderedDict

import torch

from .. import hip_autograd as A
from ..archs import build_network
from ..losses import build_loss
from ..utils.registry import MODEL_REGISTRY
from .sr_model import SRModel


@MODEL_REGISTRY.register()
class SRGANModel(SRModel):

    def init_training_settings(self):
        train_opt = self.opt['train']
        self._init_ema(train_opt)
        self.net_d = self.model_to_device(build_network(self.opt['network_d']))
        self.print_network(self.net_d)
        load_path = self.opt['path'].get('pretrain_network_d', None)
        if load_path is not None:
            self.load_network(self.net_d, load_path, self.opt['path'].get('strict_load_d', True))
        self.net_g.train()
        self.net_d.train()
        self.cri_pix = build_loss(train_opt['pixel_opt']).to(self.device) if train_opt.get('pixel_opt') else None
        # perceptual loss (losses.py:249-356 on HIP VGG features; the frozen VGG is not optimised)
        self.cri_perceptual = (build_loss(train_opt['perceptual_opt']).to(self.device)
                               if train_opt.get('perceptual_opt') else None)
        if train_opt.get('gan_opt'):
            self.cri_gan = build_loss(train_opt['gan_opt']).to(self.device)
        self.net_d_iters = train_opt.get('net_d_iters', 1)
        self.net_d_init_iters = train_opt.get('net_d_init_iters', 0)
        self.setup_optimizers()
        self.setup_schedulers()
        self._finish_ema()

    def setup_optimizers(self):
        train_opt = self.opt['train']
        optim_type = train_opt['optim_g'].pop('type')
        self.optimizer_g = self.get_optimizer(optim_type, self.net_g.parameters(), modules=[self.net_g],
                                              **train_opt['optim_g'])
        self.optimizers.append(self.optimizer_g)
        optim_type = train_opt['optim_d'].pop('type')
        self.optimizer_d = self.get_optimizer(optim_type, self.net_d.parameters(), modules=[self.net_d],
                                              **train_opt['optim_d'])
        self.optimizers.append(self.optimizer_d)

    def _g_active(self, current_iter):
        return current_iter % self.net_d_iters == 0 and current_iter > self.net_d_init_iters

    def optimize_parameters(self, current_iter):
        for p in self.net_d.parameters():
            p.requires_grad = False
        self.optimizer_g.zero_grad()
        self.output = self.net_g(self.lq)
        loss_dict = OrderedDict()
        if self._g_active(current_iter):
            l_g_total = 0
            if self.cri_pix:
                l_g_pix = self.cri_pix(self.output, self.gt)
                l_g_total = l_g_total + l_g_pix
                loss_dict['l_g_pix'] = l_g_pix
            if self.cri_perceptual:  # sr(gan)_model.py: perceptual (and style) terms of the generator loss
                l_g_percep, l_g_style = self.cri_perceptual(self.output, self.gt)
                if l_g_percep is not None:
                    l_g_total = l_g_total + l_g_percep
                    loss_dict['l_g_percep'] = l_g_percep
                if l_g_style is not None:
                    l_g_total = l_g_total + l_g_style
                    loss_dict['l_g_style'] = l_g_style
            fake_g_pred = self.net_d(self.output)
            l_g_gan = self.cri_gan(fake_g_pred, True, is_disc=False)
            l_g_total = l_g_total + l_g_gan
            loss_dict['l_g_gan'] = l_g_gan
            l_g_total.backward()
            self._step(self.optimizer_g)

        for p in self.net_d.parameters():
            p.requires_grad = True
        self.optimizer_d.zero_grad()
        real_d_pred = self.net_d(self.gt)
        l_d_real = self.cri_gan(real_d_pred, True, is_disc=True)
        loss_dict['l_d_real'] = l_d_real
        loss_dict['out_d_real'] = A.mean(real_d_pred)
        l_d_real.backward()
        fake_d_pred = self.net_d(self.output.detach())
        l_d_fake = self.cri_gan(fake_d_pred, False, is_disc=True)
        loss_dict['l_d_fake'] = l_d_fake
        loss_dict['out_d_fake'] = A.mean(fake_d_pred)
        l_d_fake.backward()
        self._step(self.optimizer_d)
        self.log_dict = self.reduce_loss_dict(loss_dict)
        if self.ema_decay > 0:
            self.model_ema(decay=self.ema_decay)

    def save(self, epoch, current_iter):
        if hasattr(self, 'net_g_ema'):
            self.save_network([self.net_g, self.net_g_ema], 'net_g', current_iter, param_key=['params', 'params_ema'])
        else:
            self.save_network(self.net_g, 'net_g', current_iter)
        self.save_network(self.net_d, 'net_d', current_iter)
        self.save_training_state(epoch, current_iter)

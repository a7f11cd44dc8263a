"""ESRGANModel: the relativistic-average GAN step of the path.

Counterpart of basicsr/models/esrgan_model.py:12-83.  Per step: 1 G forward, 1 G backward, 5 D forwards,
3 D backwards (one of them flows into G), in the reference's order and with its detach placements; the two
D backward calls accumulate into the same gradient arena, which is all-reduced ONCE before optimizer_d.step()
(mathematically identical to DDP's two all-reduces)."""
from collections import OrderedDict

import torch

from .. import hip_autograd as A
from ..utils.registry import MODEL_REGISTRY
from .srgan_model import SRGANModel


@MODEL_REGISTRY.register()
class ESRGANModel(SRGANModel):

    def optimize_parameters(self, current_iter):
        # ---- optimize net_g (D frozen: esrgan_model.py:14-15)
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
            # relativistic gan: real logits are detached (:38), fake logits carry the graph into G (:39)
            with torch.no_grad():
                real_d_pred = self.net_d(self.gt)
            fake_g_pred = self.net_d(self.output)
            l_g_real = self.cri_gan.relativistic(real_d_pred, fake_g_pred, False, is_disc=False)
            l_g_fake = self.cri_gan.relativistic(fake_g_pred, real_d_pred, True, is_disc=False)
            l_g_gan = (l_g_real + l_g_fake) / 2
            l_g_total = l_g_total + l_g_gan
            loss_dict['l_g_gan'] = l_g_gan
            l_g_total.backward()
            self._step(self.optimizer_g)

        # ---- optimize net_d (:51-73): real and fake are back-propagated separately, means are detached
        for p in self.net_d.parameters():
            p.requires_grad = True
        self.optimizer_d.zero_grad()
        with torch.no_grad():  # == self.net_d(self.output).detach() (:65): BN statistics still update
            fake_d_pred = self.net_d(self.output.detach())
        real_d_pred = self.net_d(self.gt)
        l_d_real = self.cri_gan.relativistic(real_d_pred, fake_d_pred, True, is_disc=True) * 0.5
        l_d_real.backward()
        fake_d_pred = self.net_d(self.output.detach())
        l_d_fake = self.cri_gan.relativistic(fake_d_pred, real_d_pred.detach(), False, is_disc=True) * 0.5
        l_d_fake.backward()
        self._step(self.optimizer_d)

        loss_dict['l_d_real'] = l_d_real
        loss_dict['l_d_fake'] = l_d_fake
        loss_dict['out_d_real'] = A.mean(real_d_pred)
        loss_dict['out_d_fake'] = A.mean(fake_d_pred)
        self.log_dict = self.reduce_loss_dict(loss_dict)
        if self.ema_decay > 0:
            self.model_ema(decay=self.ema_decay)

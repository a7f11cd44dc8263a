"""Losses of the ESRGAN path as HIP reductions.

``L1Loss`` mirrors basicsr/losses/losses.py:80-106 (+ weighted_loss, loss_util.py:57-95); ``GANLoss`` mirrors
:359-461 for ``gan_type='vanilla'`` (BCEWithLogitsLoss :379-380).  ``GANLoss.relativistic`` is the fused
form of ``cri_gan(a - torch.mean(b), target, is_disc)`` used by ESRGANModel (esrgan_model.py:40-41,67,71).
"""
import torch
from torch import nn

from .. import hip_autograd as A
from ..utils.registry import LOSS_REGISTRY

_reduction_modes = ['none', 'mean', 'sum']


@LOSS_REGISTRY.register()
class L1Loss(nn.Module):
    """loss_weight * mean(|pred - target|); 'sum' rescales the same reduction; element weights / 'none' are
    not on the path (the reference's ESRGAN recipes use the default 'mean', train_ESRGAN_x4.yml:84-87)."""

    def __init__(self, loss_weight=1.0, reduction='mean'):
        super().__init__()
        if reduction not in _reduction_modes:
            raise ValueError(f'Unsupported reduction mode: {reduction}. Supported ones are: {_reduction_modes}')
        self.loss_weight, self.reduction = loss_weight, reduction

    def forward(self, pred, target, weight=None, **kwargs):
        if weight is not None or self.reduction == 'none':
            raise NotImplementedError('element-wise weights / reduction="none" are not implemented on the HIP path')
        assert pred.shape == target.shape
        scale = self.loss_weight * (pred.numel() if self.reduction == 'sum' else 1.0)
        return A.L1LossFn.apply(pred, target.detach(), float(scale))


@LOSS_REGISTRY.register()
class MSELoss(L1Loss):
    """loss_weight * mean((pred - target)^2) (losses.py:165-191)."""

    def forward(self, pred, target, weight=None, **kwargs):
        if weight is not None or self.reduction == 'none':
            raise NotImplementedError('element-wise weights / reduction="none" are not implemented on the HIP path')
        assert pred.shape == target.shape
        scale = self.loss_weight * (pred.numel() if self.reduction == 'sum' else 1.0)
        return A.PixelLossFn.apply(pred, target.detach(), float(scale), 1, 0.0)


@LOSS_REGISTRY.register()
class CharbonnierLoss(L1Loss):
    """loss_weight * mean(sqrt((pred - target)^2 + eps)) (losses.py:194-227)."""

    def __init__(self, loss_weight=1.0, reduction='mean', eps=1e-12):
        super().__init__(loss_weight, reduction)
        self.eps = eps

    def forward(self, pred, target, weight=None, **kwargs):
        if weight is not None or self.reduction == 'none':
            raise NotImplementedError('element-wise weights / reduction="none" are not implemented on the HIP path')
        assert pred.shape == target.shape
        scale = self.loss_weight * (pred.numel() if self.reduction == 'sum' else 1.0)
        return A.PixelLossFn.apply(pred, target.detach(), float(scale), 2, float(self.eps))


@LOSS_REGISTRY.register()
class GANLoss(nn.Module):
    """GANLoss(gan_type, real_label_val=1.0, fake_label_val=0.0, loss_weight) with gan_type 'vanilla' | 'lsgan' | 'wgan' |
    'wgan_softplus' | 'hinge' (losses.py:360-461); loss_weight applies to generator calls only (is_disc=False), exactly as
    losses.py:460-461.  'vanilla' with the hard labels 1 / 0 — every recipe of the reference — runs the fused relativistic
    BCE reduction; the other criteria are one point-wise reduction each (sr_gan_point_loss_*)."""

    _TYPES = ('vanilla', 'lsgan', 'wgan', 'wgan_softplus', 'hinge')

    def __init__(self, gan_type, real_label_val=1.0, fake_label_val=0.0, loss_weight=1.0):
        super().__init__()
        if gan_type not in self._TYPES:
            raise NotImplementedError(f'GAN type {gan_type} is not implemented.')
        self.gan_type, self.loss_weight = gan_type, loss_weight
        self.real_label_val, self.fake_label_val = real_label_val, fake_label_val
        self._hard = gan_type == 'vanilla' and real_label_val == 1.0 and fake_label_val == 0.0

    def forward(self, input, target_is_real, is_disc=False):
        w = float(1.0 if is_disc else self.loss_weight)
        real = bool(target_is_real)
        if self._hard:
            return A.BCELogitsFn.apply(input, None, real, w)
        label = float(self.real_label_val if real else self.fake_label_val)
        sign = -1.0 if real else 1.0
        if self.gan_type == 'vanilla':        # BCE-with-logits against a soft label
            return A.GanPointLossFn.apply(input, 5, label, w)
        if self.gan_type == 'lsgan':
            return A.GanPointLossFn.apply(input, 1, label, w)
        if self.gan_type == 'wgan':
            return A.GanPointLossFn.apply(input, 2, sign, w)
        if self.gan_type == 'wgan_softplus':
            return A.GanPointLossFn.apply(input, 3, sign, w)
        # hinge: the discriminator sees relu(1 -/+ x), the generator -mean(x) whatever the target (losses.py:450-455)
        return A.GanPointLossFn.apply(input, 4, sign, w) if is_disc else A.GanPointLossFn.apply(input, 2, -1.0, w)

    def relativistic(self, pred, other, target_is_real, is_disc=False):
        """== self(pred - torch.mean(other), target_is_real, is_disc); one fused reduction for the default criterion."""
        if self._hard:
            return A.BCELogitsFn.apply(pred, other, bool(target_is_real), float(1.0 if is_disc else self.loss_weight))
        return self(pred - other.float().mean(), target_is_real, is_disc)

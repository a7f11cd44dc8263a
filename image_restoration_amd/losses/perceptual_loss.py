"""Perceptual loss on the HIP path.

Counterpart of basicsr/losses/losses.py:249-356 with the reference's constructor: ``PerceptualLoss(layer_weights,
vgg_type='vgg19', use_input_norm=True, range_norm=False, perceptual_weight=1.0, style_weight=0., criterion='l1')``;
``forward(x, gt) -> (percep_loss | None, style_loss | None)``.  Features come from archs/vgg_arch.py (HIP convolutions and
pooling), the 'l1' criterion is the HIP L1 reduction.  The Gram matrices of the style term are HIP kernels too
(``hip_autograd.GramFn``: sr_gram_fwd_f32 / sr_gram_bwd_f32) and the 'fro' criterion is the square root of the fused
squared-error reduction (``PixelLossFn``).  'l2' raises AttributeError in the reference itself
(torch.nn.L2loss does not exist, losses.py:290) and NotImplementedError here."""
import torch
from torch import nn

from ..archs.vgg_arch import VGGFeatureExtractor
from ..hip_autograd import GramFn, PixelLossFn
from ..utils.registry import LOSS_REGISTRY
from .losses import L1Loss


@LOSS_REGISTRY.register()
class PerceptualLoss(nn.Module):

    def __init__(self, layer_weights, vgg_type='vgg19', use_input_norm=True, range_norm=False, perceptual_weight=1.0,
                 style_weight=0., criterion='l1', compute_dtype='fp32', weights_path=None,
                 allow_random_init=False):
        super().__init__()
        self.perceptual_weight, self.style_weight, self.layer_weights = perceptual_weight, style_weight, dict(layer_weights)
        self.vgg = VGGFeatureExtractor(layer_name_list=list(self.layer_weights.keys()), vgg_type=vgg_type,
                                       use_input_norm=use_input_norm, range_norm=range_norm, compute_dtype=compute_dtype,
                                       weights_path=weights_path, allow_random_init=allow_random_init)
        self.criterion_type = criterion
        if criterion == 'l1':
            self.criterion = L1Loss()
        elif criterion == 'fro':
            self.criterion = None
        else:  # 'l2' included: the reference's branch for it cannot run
            raise NotImplementedError(f'{criterion} criterion has not been supported.')

    def _distance(self, a, b):
        if self.criterion_type == 'fro':  # ||a - b||_F = sqrt(numel * mean((a - b)^2))
            return torch.sqrt(PixelLossFn.apply(a.contiguous(), b.contiguous(), float(a.numel()), 1, 0.0))
        return self.criterion(a.contiguous(), b.contiguous())

    @staticmethod
    def _gram_mat(x):
        """[n, c, h, w] -> [n, c, c] = F F^T / (c h w) (losses.py:342-356)."""
        return GramFn.apply(x)

    def forward(self, x, gt):
        x_features = self.vgg(x)
        gt_features = self.vgg(gt.detach())
        if self.perceptual_weight > 0:
            percep_loss = 0
            for k in x_features.keys():
                percep_loss = percep_loss + self._distance(x_features[k], gt_features[k].detach()) * self.layer_weights[k]
            percep_loss = percep_loss * self.perceptual_weight
        else:
            percep_loss = None
        if self.style_weight > 0:
            style_loss = 0
            for k in x_features.keys():
                style_loss = style_loss + self._distance(self._gram_mat(x_features[k]),
                                                         self._gram_mat(gt_features[k].detach())) * self.layer_weights[k]
            style_loss = style_loss * self.style_weight
        else:
            style_loss = None
        return percep_loss, style_loss

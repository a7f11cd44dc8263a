"""Perceptual loss on the HIP path.

Counterpart of basicsr/losses/losses.py:249-356 with the reference's constructor: ``PerceptualLoss(layer_weights,
vgg_type='vgg19', use_input_norm=True, range_norm=False, perceptual_weight=1.0, style_weight=0., criterion='l1')``;
``forward(x, gt) -> (percep_loss | None, style_loss | None)``.  Features come from archs/vgg_arch.py (HIP convolutions and
pooling), the criterion is the HIP L1 reduction.  The Gram-matrix style term and the 'fro' criterion are not on the path
(style_weight is 0 in the reference's ESRGAN recipe, train_ESRGAN_x4.yml:88-97); 'l2' raises AttributeError in the reference
itself (torch.nn.L2loss does not exist, losses.py:290)."""
from torch import nn

from ..archs.vgg_arch import VGGFeatureExtractor
from ..utils.registry import LOSS_REGISTRY
from .losses import L1Loss


@LOSS_REGISTRY.register()
class PerceptualLoss(nn.Module):

    def __init__(self, layer_weights, vgg_type='vgg19', use_input_norm=True, range_norm=False, perceptual_weight=1.0,
                 style_weight=0., criterion='l1'):
        super().__init__()
        self.perceptual_weight, self.style_weight, self.layer_weights = perceptual_weight, style_weight, dict(layer_weights)
        self.vgg = VGGFeatureExtractor(layer_name_list=list(self.layer_weights.keys()), vgg_type=vgg_type,
                                       use_input_norm=use_input_norm, range_norm=range_norm)
        self.criterion_type = criterion
        if criterion == 'l1':
            self.criterion = L1Loss()
        elif criterion in ('l2', 'fro'):
            raise NotImplementedError(f"criterion '{criterion}' is not on the HIP path (the reference's 'l2' does not run either)")
        else:
            raise NotImplementedError(f'{criterion} criterion has not been supported.')
        if style_weight > 0:
            raise NotImplementedError('the Gram-matrix style loss is not on the HIP path')

    def forward(self, x, gt):
        x_features = self.vgg(x)
        gt_features = self.vgg(gt.detach())
        if self.perceptual_weight > 0:
            percep_loss = 0
            for k in x_features.keys():
                percep_loss = percep_loss + self.criterion(x_features[k], gt_features[k].detach()) * self.layer_weights[k]
            percep_loss = percep_loss * self.perceptual_weight
        else:
            percep_loss = None
        return percep_loss, None

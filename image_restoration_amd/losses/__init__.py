"""Loss registry + factory (reference: basicsr/losses/__init__.py:14-26)."""
import logging
from copy import deepcopy

from ..utils.registry import LOSS_REGISTRY
from .losses import CharbonnierLoss, GANLoss, L1Loss, MSELoss  # noqa: F401
from .perceptual_loss import PerceptualLoss  # noqa: F401

__all__ = ['build_loss', 'L1Loss', 'MSELoss', 'CharbonnierLoss', 'GANLoss', 'PerceptualLoss']


def build_loss(opt):
    opt = deepcopy(opt)
    loss_type = opt.pop('type')
    loss = LOSS_REGISTRY.get(loss_type)(**opt)
    logging.getLogger('basicsr').info(f'Loss [{loss.__class__.__name__}] is created.')
    return loss

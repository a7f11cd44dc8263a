"""Losses of the path and their factory; every class files itself in LOSS_REGISTRY when its module is imported
(counterpart of basicsr/losses/__init__.py:14-26)."""
from ..utils.registry import LOSS_REGISTRY, instantiate
from .losses import CharbonnierLoss, GANLoss, L1Loss, MSELoss  # noqa: F401
from .perceptual_loss import PerceptualLoss  # noqa: F401

__all__ = ['build_loss', 'L1Loss', 'MSELoss', 'CharbonnierLoss', 'GANLoss', 'PerceptualLoss']


def build_loss(opt):
    """``{type: L1Loss, loss_weight: ..}`` -> loss module."""
    return instantiate(LOSS_REGISTRY, opt, 'Loss')

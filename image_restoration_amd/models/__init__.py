"""Model registry + factory (reference: basicsr/models/__init__.py:19-30): ``build_model(opt)`` picks
``opt['model_type']`` from MODEL_REGISTRY."""
import logging
from copy import deepcopy

from ..utils.registry import MODEL_REGISTRY
from .sr_model import SRModel  # noqa: F401
from .srgan_model import SRGANModel  # noqa: F401
from .esrgan_model import ESRGANModel  # noqa: F401

__all__ = ['build_model']


def build_model(opt):
    opt = deepcopy(opt)
    model = MODEL_REGISTRY.get(opt['model_type'])(opt)
    logging.getLogger('basicsr').info(f'Model [{model.__class__.__name__}] is created.')
    return model

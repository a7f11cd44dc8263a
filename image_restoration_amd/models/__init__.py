"""Model registry + factory (reference: basicsr/models/__init__.py:19-30): ``build_model(opt)`` picks
``opt['model_type']`` from MODEL_REGISTRY."""
from ..utils.registry import MODEL_REGISTRY, instantiate
from .sr_model import SRModel  # noqa: F401
from .srgan_model import SRGANModel  # noqa: F401
from .esrgan_model import ESRGANModel  # noqa: F401


def build_model(opt):
    """The whole option dict -> model of class ``opt['model_type']``."""
    return instantiate(MODEL_REGISTRY, opt, 'Model', type_key='model_type', as_kwargs=False)

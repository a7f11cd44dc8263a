"""ESRGANModel: SRGANModel's step driver with the relativistic average GAN terms switched on.

Behavioural counterpart of basicsr/models/esrgan_model.py:12-83.  Per step: 1 G forward, 1 G backward, 5 D forwards
(two in the generator phase, three in the critic phase, one of those without a graph), 3 D backwards of which one
flows into G.  The terms themselves are in models/srgan_model.py (``relativistic`` branches), pinned by golden G-i."""
from ..utils.registry import MODEL_REGISTRY
from .srgan_model import SRGANModel


@MODEL_REGISTRY.register()
class ESRGANModel(SRGANModel):
    relativistic = True

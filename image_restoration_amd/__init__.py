"""MI355X-native RRDBNet / ESRGAN x4 super-resolution hot path.

Drop-in for the BasicSR-derived ``basicsr.archs`` registry surface of the reference
(ChuRuaNh0/Image_Restoration, Car_Plate-Restoration/basicsr): ``ARCH_REGISTRY``,
``build_network(opt)`` and ``RRDBNet`` keep their names, constructor arguments and
state_dict keys; the arithmetic runs in hand-written gfx950 HIP kernels behind the C ABI
of ``include/sr_hip.h``.
"""
from .utils.registry import ARCH_REGISTRY, LOSS_REGISTRY, MODEL_REGISTRY, DATASET_REGISTRY, METRIC_REGISTRY  # noqa: F401
from .archs import build_network  # noqa: F401

__version__ = '0.1.0'

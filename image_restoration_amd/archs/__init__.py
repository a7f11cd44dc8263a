"""Architecture registry + factory (reference: basicsr/archs/__init__.py:13-25).

Every ``*_arch.py`` in this folder is imported at package import so its classes register
themselves; ``build_network(opt)`` pops ``type`` and forwards the remaining keys as kwargs.
"""
import importlib
import os

from ..utils.registry import ARCH_REGISTRY, instantiate

__all__ = ['build_network']

_folder = os.path.dirname(os.path.abspath(__file__))
_arch_modules = [
    importlib.import_module(f'{__name__}.{f[:-3]}') for f in sorted(os.listdir(_folder)) if f.endswith('_arch.py')
]


def build_network(opt):
    """``{type: RRDBNet, num_in_ch: 3, ..}`` -> nn.Module."""
    return instantiate(ARCH_REGISTRY, opt, 'Network')

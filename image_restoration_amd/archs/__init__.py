"""Architecture registry + factory (reference: basicsr/archs/__init__.py:13-25).

Every ``*_arch.py`` in this folder is imported at package import so its classes register
themselves; ``build_network(opt)`` pops ``type`` and forwards the remaining keys as kwargs.
"""
import importlib
import logging
import os
from copy import deepcopy

from ..utils.registry import ARCH_REGISTRY

__all__ = ['build_network']

_folder = os.path.dirname(os.path.abspath(__file__))
_arch_modules = [
    importlib.import_module(f'{__name__}.{f[:-3]}') for f in sorted(os.listdir(_folder)) if f.endswith('_arch.py')
]


def build_network(opt):
    opt = deepcopy(opt)
    network_type = opt.pop('type')
    net = ARCH_REGISTRY.get(network_type)(**opt)
    logging.getLogger('basicsr').info(f'Network [{net.__class__.__name__}] is created.')
    return net

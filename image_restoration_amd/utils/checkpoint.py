"""Checkpoint key maps.

The officially released ESRGAN weights name the generator's modules differently from BasicSR; the reference ships a
one-off converter (scripts/model_conversion/convert_models.py:174-199: rdb->RDB, body->RRDB_trunk,
conv_body->trunk_conv, conv_up->upconv, conv_hr->HRconv).  Here the same map is a function, applied automatically
by ``load_generator_weights`` so either naming loads into the HIP RRDBNet.
"""
from collections import OrderedDict

import torch


def basicsr_to_official_key(k):
    if 'rdb' in k:
        return k.replace('rdb', 'RDB').replace('body', 'RRDB_trunk')
    if 'conv_body' in k:
        return k.replace('conv_body', 'trunk_conv')
    if 'conv_up' in k:
        return k.replace('conv_up', 'upconv')
    if 'conv_hr' in k:
        return k.replace('conv_hr', 'HRconv')
    return k


def official_to_basicsr_key(k):
    if 'RRDB_trunk' in k:
        return k.replace('RRDB_trunk', 'body').replace('RDB', 'rdb')
    if 'trunk_conv' in k:
        return k.replace('trunk_conv', 'conv_body')
    if 'upconv' in k:
        return k.replace('upconv', 'conv_up')
    if 'HRconv' in k:
        return k.replace('HRconv', 'conv_hr')
    return k


def is_official_esrgan(state_dict):
    return any('RRDB_trunk' in k or 'trunk_conv' in k or 'HRconv' in k for k in state_dict)


def convert_official_esrgan(state_dict):
    """Official ESRGAN generator state_dict -> BasicSR / this package's keys."""
    return OrderedDict((official_to_basicsr_key(k[7:] if k.startswith('module.') else k), v) for k, v in state_dict.items())


def load_generator_weights(net, path_or_state, strict=True, prefer='params_ema'):
    """Loads ``{'params'|'params_ema': sd}`` files (base_model.py:251-277), bare state_dicts and official ESRGAN files."""
    sd = torch.load(path_or_state, map_location='cpu', weights_only=False) if isinstance(path_or_state, str) else path_or_state
    if isinstance(sd, dict) and ('params' in sd or 'params_ema' in sd):
        sd = sd.get(prefer, sd.get('params', sd.get('params_ema')))
    sd = OrderedDict((k[7:] if k.startswith('module.') else k, v) for k, v in sd.items())
    if is_official_esrgan(sd):
        sd = convert_official_esrgan(sd)
    out = net.load_state_dict(sd, strict=strict)
    if hasattr(net, 'invalidate_packed'):
        net.invalidate_packed()
    return out

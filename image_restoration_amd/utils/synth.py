"""Deterministic synthetic parameters and inputs (numpy PCG64: stable across versions).

Never relies on torch's RNG streams, so the same seed gives the same network here, in the
golden generator (which loads them into the REFERENCE modules) and on the GPU box.
RDB conv weights ~ N(0, (0.1*sqrt(2/fan_in))^2) like the reference initialisation
(arch_util.py:12-40 with scale 0.1), other convs ~ U(+-1/sqrt(fan_in)); biases are small and
NON-zero so bias paths are exercised.
"""
import math
from collections import OrderedDict

import numpy as np


def rrdbnet_param_shapes(num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32):
    """(name, shape) in state_dict order of RRDBNet (rrdbnet_arch.py:94-101)."""
    cin = num_in_ch * {4: 1, 2: 4, 1: 16}.get(scale, 1)
    out = []

    def conv(name, ci, co):
        out.append((f'{name}.weight', (co, ci, 3, 3)))
        out.append((f'{name}.bias', (co,)))

    conv('conv_first', cin, num_feat)
    for b in range(num_block):
        for r in (1, 2, 3):
            for k in range(1, 5):
                conv(f'body.{b}.rdb{r}.conv{k}', num_feat + (k - 1) * num_grow_ch, num_grow_ch)
            conv(f'body.{b}.rdb{r}.conv5', num_feat + 4 * num_grow_ch, num_feat)
    for name in ('conv_body', 'conv_up1', 'conv_up2', 'conv_hr'):
        conv(name, num_feat, num_feat)
    conv('conv_last', num_feat, num_out_ch)
    return out


def conv_params(rng, shape_w, rdb_style, bias_scale=0.05):
    co, ci, kh, kw = shape_w
    fan_in = ci * kh * kw
    if rdb_style:
        w = rng.standard_normal(shape_w, dtype=np.float32) * np.float32(0.1 * math.sqrt(2.0 / fan_in))
    else:
        bound = 1.0 / math.sqrt(fan_in)
        w = (rng.random(shape_w, dtype=np.float32) * 2 - 1) * np.float32(bound)
    b = (rng.random((co,), dtype=np.float32) * 2 - 1) * np.float32(bias_scale)
    return w.astype(np.float32), b.astype(np.float32)


def rrdbnet_state_dict(seed=0, **cfg):
    """OrderedDict name -> np.float32 array for RRDBNet(**cfg)."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    shapes = rrdbnet_param_shapes(**cfg)
    for i in range(0, len(shapes), 2):
        (wn, ws), (bn, _) = shapes[i], shapes[i + 1]
        w, b = conv_params(rng, ws, rdb_style='.rdb' in wn)
        sd[wn], sd[bn] = w, b
    return sd


def rdb_state_dict(seed, num_feat=64, num_grow_ch=32, prefix=''):
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for k in range(1, 6):
        ci = num_feat + (k - 1) * num_grow_ch
        co = num_grow_ch if k < 5 else num_feat
        w, b = conv_params(rng, (co, ci, 3, 3), rdb_style=True)
        sd[f'{prefix}conv{k}.weight'], sd[f'{prefix}conv{k}.bias'] = w, b
    return sd


def rrdb_state_dict(seed, num_feat=64, num_grow_ch=32):
    sd = OrderedDict()
    for r in (1, 2, 3):
        sd.update(rdb_state_dict(seed * 10 + r, num_feat, num_grow_ch, prefix=f'rdb{r}.'))
    return sd


def uniform_input(seed, shape):
    """U[0,1) fp32 images, the benchmark's synthetic input."""
    return np.random.default_rng(seed).random(shape, dtype=np.float32)


def signed_input(seed, shape, scale=1.0):
    """Zero-mean features for block-level tests."""
    return ((np.random.default_rng(seed).random(shape, dtype=np.float32) * 2 - 1) * np.float32(scale)).astype(np.float32)


def vgg128_param_shapes(num_in_ch, num_feat, input_size=128):
    """(name, shape, kind) in state_dict order of VGGStyleDiscriminator128 (discriminator_arch.py:21-46) or, with
    input_size=256, VGGStyleDiscriminator256 (:75-120: one more 8nf -> 8nf stage)."""
    nf = num_feat
    out = []

    def conv(name, ci, co, k, bias):
        out.append((f'{name}.weight', (co, ci, k, k), 'conv'))
        if bias:
            out.append((f'{name}.bias', (co,), 'bias'))

    def bn(name, c):
        out.extend([(f'{name}.weight', (c,), 'bn_w'), (f'{name}.bias', (c,), 'bias'), (f'{name}.running_mean', (c,), 'rm'),
                    (f'{name}.running_var', (c,), 'rv'), (f'{name}.num_batches_tracked', (), 'nbt')])

    conv('conv0_0', num_in_ch, nf, 3, True)
    conv('conv0_1', nf, nf, 4, False)
    bn('bn0_1', nf)
    widths = [(nf, nf * 2), (nf * 2, nf * 4), (nf * 4, nf * 8), (nf * 8, nf * 8)] + ([(nf * 8, nf * 8)] if input_size == 256 else [])
    for i, (ci, co) in enumerate(widths, start=1):
        conv(f'conv{i}_0', ci, co, 3, False)
        bn(f'bn{i}_0', co)
        conv(f'conv{i}_1', co, co, 4, False)
        bn(f'bn{i}_1', co)
    out.extend([('linear1.weight', (100, nf * 8 * 16), 'lin'), ('linear1.bias', (100,), 'bias'),
                ('linear2.weight', (1, 100), 'lin'), ('linear2.bias', (1,), 'bias')])
    return out


def vgg128_state_dict(seed, num_in_ch=3, num_feat=64, input_size=128):
    """Deterministic VGGStyleDiscriminator128 / 256 state (non-trivial BN affine and running statistics)."""
    rng = np.random.default_rng(seed)
    sd = OrderedDict()
    for name, shape, kind in vgg128_param_shapes(num_in_ch, num_feat, input_size):
        if kind == 'conv':
            fan_in = shape[1] * shape[2] * shape[3]
            sd[name] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(math.sqrt(2.0 / fan_in) * 0.7)).astype(np.float32)
        elif kind == 'lin':
            sd[name] = ((rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(1.0 / math.sqrt(shape[1]))).astype(np.float32)
        elif kind == 'bias':
            sd[name] = ((rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(0.1)).astype(np.float32)
        elif kind == 'bn_w':
            sd[name] = (1.0 + (rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(0.2)).astype(np.float32)
        elif kind == 'rm':
            sd[name] = ((rng.random(shape, dtype=np.float32) * 2 - 1) * np.float32(0.1)).astype(np.float32)
        elif kind == 'rv':
            sd[name] = (0.5 + rng.random(shape, dtype=np.float32)).astype(np.float32)
        else:
            sd[name] = np.array(0, dtype=np.int64)
    return sd


def smooth_pairs(seed, n, size, scale=4):
    """(lq, gt) batches with learnable structure for end-to-end training checks: gt = bicubic x8 enlargement of uniform
    noise (smooth colour fields in [0, 1], [n, 3, size, size]), lq = its scale x scale box average."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(n, 3, size // 8 + 2, size // 8 + 2, generator=g)
    gt = F.interpolate(low, scale_factor=8, mode='bicubic', align_corners=False)[:, :, 8:8 + size, 8:8 + size].clamp(0, 1)
    return F.avg_pool2d(gt, scale), gt.contiguous()

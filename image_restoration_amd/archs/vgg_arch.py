"""VGG feature extractor of the perceptual loss on the HIP path.

Counterpart of basicsr/archs/vgg_arch.py:55-162: ``VGGFeatureExtractor(layer_name_list, vgg_type='vgg19',
use_input_norm=True, range_norm=False, requires_grad=False, remove_pooling=False, pooling_stride=2)`` returns
``{layer name: feature}`` for the requested layers of torchvision's VGG11/13/16/19 ``features`` stack.  The reference
takes that stack (and its ImageNet weights) from torchvision, which is not installed here and cannot be downloaded:
the layer tables below restate torchvision's configurations A/B/D/E, the parameters are created with torchvision's
initialisation, and pretrained weights are loaded when a file is present (the reference's VGG_PRETRAIN_PATH, or
``weights_path``) — a torchvision ``features.N.weight`` state_dict or this module's own keys.  **Parity unpinned**:
with neither torchvision nor its weights available, tests compare against a PyTorch-CPU restatement
(oracle/vgg_ref.py) with the same random weights.

Every layer is libsr_hip.so launches: 3x3 conv (+ReLU fused as LeakyReLU slope 0), 2x2 max-pool, input normalisation.
BatchNorm variants ('vgg19_bn', ...) are not on the path.
"""
import logging
import os

import torch
from torch import nn
from torch.nn import init

from .. import _lib
from .. import hip_autograd as A
from ..utils.registry import ARCH_REGISTRY
from .arch_util import Conv3x3Params

VGG_PRETRAIN_PATH = 'experiments/pretrained_models/vgg19-dcbb9e9d.pth'  # the reference's location (vgg_arch.py:10)

# output channels per stage and convs per stage of torchvision's configurations A, B, D, E
_STAGES = {'vgg11': (1, 1, 2, 2, 2), 'vgg13': (2, 2, 2, 2, 2), 'vgg16': (2, 2, 3, 3, 3), 'vgg19': (2, 2, 4, 4, 4)}
_WIDTHS = (64, 128, 256, 512, 512)


def layer_names(vgg_type):
    """['conv1_1', 'relu1_1', ..., 'pool5'] in torchvision's ``features`` order (the reference's NAMES table)."""
    names = []
    for stage, nconv in enumerate(_STAGES[vgg_type], start=1):
        for k in range(1, nconv + 1):
            names += [f'conv{stage}_{k}', f'relu{stage}_{k}']
        names.append(f'pool{stage}')
    return names


@ARCH_REGISTRY.register()
class VGGFeatureExtractor(nn.Module):

    def __init__(self, layer_name_list, vgg_type='vgg19', use_input_norm=True, range_norm=False, requires_grad=False,
                 remove_pooling=False, pooling_stride=2, weights_path=None, compute_dtype='fp32', allow_random_init=False):
        super().__init__()
        if compute_dtype not in ('fp32', 'bf16'):
            raise ValueError(f"compute dtype must be 'fp32' or 'bf16', got {compute_dtype!r}")
        self.compute_dtype = compute_dtype  # 'bf16': CB16 activations on the generator's bf16 conv kernels, fp32 features out
        if 'bn' in vgg_type:
            raise NotImplementedError('BatchNorm VGG variants are not on the HIP path')
        if pooling_stride != 2:
            raise NotImplementedError('only MaxPool2d(kernel_size=2, stride=2) is on the HIP path')
        self.layer_name_list = list(layer_name_list)
        self.use_input_norm, self.range_norm, self.remove_pooling = use_input_norm, range_norm, remove_pooling
        self.names = layer_names(vgg_type)
        max_idx = max(self.names.index(v) for v in self.layer_name_list)  # ValueError for an unknown layer, like the reference
        self.names = self.names[:max_idx + 1]
        self.vgg_net = nn.Module()  # parameters live under vgg_net.convS_K like the reference's OrderedDict Sequential
        cin = 3
        # Random features are an explicit opt-in (tests, benchmarks).  They come from a private generator with a fixed
        # seed, so every rank of a job (seeded manual_seed + rank) builds the SAME extractor and optimises the same loss.
        gen = torch.Generator().manual_seed(0x56474731)
        for name in self.names:
            if name.startswith('conv'):
                cout = _WIDTHS[int(name[4]) - 1]
                conv = Conv3x3Params(cin, cout, bias=True)
                init.kaiming_normal_(conv.weight, mode='fan_out', nonlinearity='relu', generator=gen)  # torchvision's VGG._initialize_weights
                init.zeros_(conv.bias)
                self.vgg_net.add_module(name, conv)
                cin = cout
        path = weights_path or (VGG_PRETRAIN_PATH if os.path.exists(VGG_PRETRAIN_PATH) else None)
        if path:
            self.load_pretrained(torch.load(path, map_location='cpu', weights_only=True))
        elif allow_random_init:
            logging.getLogger('basicsr').warning(
                'VGGFeatureExtractor: allow_random_init - no pretrained weights (%s absent), features are RANDOM', VGG_PRETRAIN_PATH)
        else:
            # the reference falls back to torchvision's download here; a perceptual loss on random features would train
            # silently towards nothing, so refuse instead
            raise FileNotFoundError(
                f'VGGFeatureExtractor: no pretrained weights: {VGG_PRETRAIN_PATH} is absent and no weights_path was given '
                '(torchvision is not available to fetch them).  Pass weights_path=<vgg19-dcbb9e9d.pth>, or '
                'allow_random_init=True for tests / benchmarks.')
        for p in self.parameters():
            p.requires_grad = bool(requires_grad)
        # (x [+1]/2 - mean) / std as one per-channel affine
        mean, std = torch.tensor([0.485, 0.456, 0.406]), torch.tensor([0.229, 0.224, 0.225])
        scale = torch.tensor(0.5 if range_norm else 1.0)
        shift = torch.tensor(0.5 if range_norm else 0.0)
        if use_input_norm:
            self.register_buffer('norm_a', scale / std)
            self.register_buffer('norm_b', (shift - mean) / std)
        else:
            self.register_buffer('norm_a', scale.expand(3).clone())
            self.register_buffer('norm_b', shift.expand(3).clone())

    def load_pretrained(self, sd):
        """Accepts torchvision's ``features.N.{weight,bias}`` keys (N = index in the full features stack) or own keys."""
        own = self.state_dict()
        if any(k.startswith('features.') for k in sd):
            full = layer_names(next(t for t in _STAGES if len(layer_names(t)) >= len(self.names) and layer_names(t)[:len(self.names)] == self.names))
            mapped = {}
            for idx, name in enumerate(full):
                if name in self.names and name.startswith('conv'):
                    for leaf in ('weight', 'bias'):
                        mapped[f'vgg_net.{name}.{leaf}'] = sd[f'features.{idx}.{leaf}']
            sd = mapped
        own.update({k: v for k, v in sd.items() if k in own})
        self.load_state_dict(own)

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.SrHipError('VGGFeatureExtractor.forward runs only on a HIP device (no CPU fallback)')
        if self.range_norm or self.use_input_norm:
            x = A.ChannelAffineFn.apply(x, self.norm_a, self.norm_b)
        if self.compute_dtype == 'bf16':
            from .. import hip_autograd_bf16 as B
            to_cb, from_cb, pool, lrelu = B.ToCB16.apply, B.FromCB16.apply, B.MaxPool2x2Fn16.apply, B.LReLUFn16.apply

            def convf(t, w, b, slope):
                return B.ConvFn16.apply(t, w, b, slope, False)
        else:
            to_cb, from_cb, pool, lrelu, convf = A.ToCB8.apply, A.FromCB8.apply, A.MaxPool2x2Fn.apply, A.LReLUFn.apply, A.ConvFn.apply
        feat = to_cb(x.contiguous().float())
        want = set(self.layer_name_list)
        out = {}
        i = 0
        while i < len(self.names):
            name = self.names[i]
            if name.startswith('conv'):
                conv = getattr(self.vgg_net, name)
                relu_follows = i + 1 < len(self.names)
                if name in want or not relu_follows:  # the conv output itself is a feature (e.g. conv5_4 before relu5_4)
                    feat = convf(feat, conv.weight, conv.bias, 1.0)
                    out[name] = from_cb(feat, conv.weight.size(0))
                    if relu_follows:
                        feat = lrelu(feat, 0.0)
                else:
                    feat = convf(feat, conv.weight, conv.bias, 0.0)  # conv + ReLU fused
                if relu_follows:
                    i += 1
                    if self.names[i] in want:
                        out[self.names[i]] = from_cb(feat, conv.weight.size(0))
            elif name.startswith('pool'):
                if not self.remove_pooling:
                    feat = pool(feat)
                    if name in want:
                        out[name] = from_cb(feat, _WIDTHS[int(name[4]) - 1])
            i += 1
        return {k: out[k] for k in self.layer_name_list if k in out}

"""Parameter containers and initialisers for the HIP-backed architectures.

``Conv3x3Params`` only OWNS an OIHW weight and a bias with nn.Conv2d's state_dict keys
and default initialisation; it has no forward — the arithmetic of every layer is issued
by the owning network through libsr_hip.so.
"""
import math

import torch
from torch import nn
from torch.nn import init


class Conv3x3Params(nn.Module):
    """weight [cout, cin, 3, 3] + bias [cout], initialised like nn.Conv2d(cin, cout, 3, 1, 1)
    (kaiming_uniform(a=sqrt(5)) / U(+-1/sqrt(fan_in))), which the reference keeps for the six
    non-RDB convs of RRDBNet (rrdbnet_arch.py:94-101)."""

    def __init__(self, cin, cout, bias=True, ksize=3):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = cin, cout, ksize
        self.weight = nn.Parameter(torch.empty(cout, cin, ksize, ksize))
        self.bias = nn.Parameter(torch.empty(cout)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.in_channels * self.kernel_size * self.kernel_size
            bound = 1 / math.sqrt(fan_in)
            init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride=1, padding=1 [HIP]'

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError('Conv3x3Params is a parameter container; the owning network launches the HIP kernels')


@torch.no_grad()
def default_init_weights(module_list, scale=1, bias_fill=0, **kwargs):
    """kaiming_normal_ * scale, bias = bias_fill — the RDB initialisation of the reference
    (arch_util.py:12-40, called with scale 0.1 at rrdbnet_arch.py:30)."""
    if not isinstance(module_list, list):
        module_list = [module_list]
    for module in module_list:
        for m in module.modules():
            if isinstance(m, (Conv3x3Params, nn.Conv2d, nn.Linear)):
                init.kaiming_normal_(m.weight, **kwargs)
                m.weight.data *= scale
                if m.bias is not None:
                    m.bias.data.fill_(bias_fill)
            elif isinstance(m, nn.modules.batchnorm._BatchNorm):
                init.constant_(m.weight, 1)
                if m.bias is not None:
                    m.bias.data.fill_(bias_fill)


def make_layer(basic_block, num_basic_block, **kwarg):
    """nn.Sequential of `num_basic_block` blocks (reference arch_util.py:43-56) — gives the
    ``body.{i}.`` state_dict prefix."""
    return nn.Sequential(*[basic_block(**kwarg) for _ in range(num_basic_block)])

"""U-Net discriminator with spectral normalisation on the HIP path.

BASELINE.json names ``UNetDiscriminatorSN`` for the 128->512 training configs, but the mounted reference does not
contain it (this fork predates it; ``grep -ri UNetDiscriminator /root/reference`` has no hits).
It is built here from the published architecture (Real-ESRGAN, arXiv:2107.10833 §3.3):
conv0 3x3(+bias)+LReLU -> 3x [SN conv4x4/s2, no bias, LReLU] -> 3x [bilinear x2 (align_corners=False) -> SN conv3x3,
no bias, LReLU, + skip from the matching encoder level] -> 2x SN conv3x3 + LReLU -> conv3x3 -> 1 logit map.
**Parity unpinned by the reference**: the oracle (oracle/unet_discriminator_ref.py) is a restatement of the same
published architecture with torch.nn.utils.spectral_norm; tests compare against that.

state_dict keys follow torch.nn.utils.spectral_norm: ``convK.weight_orig`` (parameter), ``convK.weight_u`` /
``convK.weight_v`` (buffers); conv0 / conv9 have ``weight`` and ``bias``.
"""
import math

import torch
from torch import nn
from torch.nn import init

from .. import _lib
from .. import hip_autograd as A
from ..utils.registry import ARCH_REGISTRY
from .arch_util import Conv3x3Params


class SNConvParams(nn.Module):
    """weight_orig [cout, cin, k, k] + power-iteration buffers weight_u [cout], weight_v [cin*k*k]."""

    def __init__(self, cin, cout, ksize, eps=1e-12):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.eps = cin, cout, ksize, eps
        w = torch.empty(cout, cin, ksize, ksize)
        init.kaiming_uniform_(w, a=math.sqrt(5))
        self.weight_orig = nn.Parameter(w)
        # torch.nn.utils.spectral_norm initialises u, v as normalised gaussians
        self.register_buffer('weight_u', nn.functional.normalize(torch.randn(cout), dim=0, eps=eps))
        self.register_buffer('weight_v', nn.functional.normalize(torch.randn(cin * ksize * ksize), dim=0, eps=eps))

    def weight(self):
        return A.SpectralNormFn.apply(self.weight_orig, self.weight_u, self.weight_v, self.training, self.eps)

    def extra_repr(self):
        return f'{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, spectral_norm [HIP]'


@ARCH_REGISTRY.register()
class UNetDiscriminatorSN(nn.Module):
    """UNetDiscriminatorSN(num_in_ch, num_feat=64, skip_connection=True): [N, C, H, W] -> [N, 1, H, W] logits
    (H, W multiples of 8)."""

    def __init__(self, num_in_ch, num_feat=64, skip_connection=True, compute_dtype='fp32'):
        super().__init__()
        nf = num_feat
        self.num_in_ch, self.num_feat, self.skip_connection = num_in_ch, nf, skip_connection
        if compute_dtype not in ('fp32', 'bf16'):
            raise ValueError(f"compute dtype must be 'fp32' or 'bf16', got {compute_dtype!r}")
        # 'bf16': activations / activation gradients in CB16 bf16 on the generator's bf16 kernels (num_feat % 16 == 0);
        # weights, spectral normalisation, weight gradients and the optimiser stay fp32
        self.compute_dtype = compute_dtype
        assert compute_dtype == 'fp32' or nf % 16 == 0, 'bf16 needs num_feat to be a multiple of 16'
        self.conv0 = Conv3x3Params(num_in_ch, nf, bias=True)
        self.conv1 = SNConvParams(nf, nf * 2, 4)
        self.conv2 = SNConvParams(nf * 2, nf * 4, 4)
        self.conv3 = SNConvParams(nf * 4, nf * 8, 4)
        self.conv4 = SNConvParams(nf * 8, nf * 4, 3)
        self.conv5 = SNConvParams(nf * 4, nf * 2, 3)
        self.conv6 = SNConvParams(nf * 2, nf, 3)
        self.conv7 = SNConvParams(nf, nf, 3)
        self.conv8 = SNConvParams(nf, nf, 3)
        self.conv9 = Conv3x3Params(nf, 1, bias=True)

    def _sn_weights(self):
        """The normalised weights of conv1..conv8 for this forward: one power iteration each in train mode, all eight layers in one
        call (five launches instead of forty).  Returns {layer index: weight}."""
        layers = [getattr(self, f'conv{i}') for i in range(1, 9)]
        flat = []
        for m in layers:
            flat += [m.weight_orig, m.weight_u, m.weight_v]
        eps = layers[0].eps
        outs = A.SpectralNormBatchFn.apply(self.training, eps, *flat)
        return {i + 1: w for i, w in enumerate(outs)}

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.SrHipError('UNetDiscriminatorSN.forward runs only on a HIP device (no CPU fallback)')
        assert x.size(2) % 8 == 0 and x.size(3) % 8 == 0, f'input {tuple(x.shape)} must be a multiple of 8 in H and W'
        if self.compute_dtype == 'bf16':
            return self._forward_driver_bf16(x) if self.use_driver else self._forward_bf16(x)
        sn = self._sn_weights()
        conv = A.ConvFn.apply
        x0 = conv(A.ToCB8.apply(x.contiguous().float()), self.conv0.weight, self.conv0.bias, 0.2)
        x1 = conv(x0, sn[1], None, 0.2)
        x2 = conv(x1, sn[2], None, 0.2)
        x3 = conv(x2, sn[3], None, 0.2)
        x3 = A.Bilinear2xFn.apply(x3)
        x4 = conv(x3, sn[4], None, 0.2)
        if self.skip_connection:
            x4 = A.AddFn.apply(x4, x2)
        x4 = A.Bilinear2xFn.apply(x4)
        x5 = conv(x4, sn[5], None, 0.2)
        if self.skip_connection:
            x5 = A.AddFn.apply(x5, x1)
        x5 = A.Bilinear2xFn.apply(x5)
        x6 = conv(x5, sn[6], None, 0.2)
        if self.skip_connection:
            x6 = A.AddFn.apply(x6, x0)
        out = conv(x6, sn[7], None, 0.2)
        out = conv(out, sn[8], None, 0.2)
        out = conv(out, self.conv9.weight, self.conv9.bias, 1.0)
        return A.FromCB8.apply(out, 1)

    # ------------------------------------------------------------------ whole-network driver (bf16)
    use_driver = True   # False: one autograd function per layer (_forward_bf16: the same launches; kept as the drivers' cross-check)

    def _cfg(self):
        return _lib.UNetCfg(self.num_in_ch, self.num_feat, int(bool(self.skip_connection)))

    def _workspace(self, nbytes, dev):
        ws = getattr(self, '_ws', None)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            ws = self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws

    def _forward_driver_bf16(self, x):
        """archs/unet_disc_autograd.py: sr_unet_forward_bf16 / sr_unet_backward_bf16 behind ONE autograd function; the spectral
        normalisation of conv1-8 stays the batched function in front of it.  Weights whose requires_grad flags differ (a partially
        frozen network) take the per-layer route."""
        if torch.is_grad_enabled() and len({p.requires_grad for p in self.parameters()}) > 1:
            return self._forward_bf16(x)
        sn = self._sn_weights()
        weights = [self.conv0.weight, self.conv0.bias] + [sn[i] for i in range(1, 9)] + [self.conv9.weight, self.conv9.bias]
        from .unet_disc_autograd import unet_apply
        return unet_apply(self, x, weights)

    def forward_layers(self, x):
        """The same network one autograd function per layer (both precisions)."""
        keep, self.use_driver = self.use_driver, False
        try:
            return self.forward(x)
        finally:
            self.use_driver = keep

    def _forward_bf16(self, x):
        """Same network on CB16 bf16 activations.  Neighbouring layers share memory passes (hip_autograd_bf16.py): the
        encoder activations exist pixel-unshuffled only and fork into (skip, strided-conv input) with one fused gradient pass, the first two skip
        additions ride on the bilinear resampling, the resampling's backward applies the LeakyReLU derivative of the conv
        that fed it, and conv8 / conv9 apply their producer's LeakyReLU derivative in the data-gradient epilogue."""
        from .. import hip_autograd_bf16 as B
        skip = self.skip_connection
        sn = self._sn_weights()
        # x0, x1, x2 are stored pixel-unshuffled ONLY (u0, u1, u2: what the next strided conv reads): their producers write that
        # layout from the epilogue and the skip consumers read it where it is — no unshuffle passes, no second copy
        def conv(t, w, b, slope, nchw=False, **kw):   # noqa: F811 - adds the out_unshuffled option
            return B.ConvFn16.apply(t, w, b, slope, nchw, kw.get('pre_unshuffled', False), kw.get('input_slope', 1.0),
                                    kw.get('grad_premasked', False), kw.get('out_unshuffled', False), kw.get('skip_u2'))
        u0 = conv(B.ToCB16.apply(x.contiguous().float()), self.conv0.weight, self.conv0.bias, 0.2, grad_premasked=True, out_unshuffled=True)
        s0, u0 = B.ForkU2Fn16.apply(u0, 0.2)
        u1 = conv(u0, sn[1], None, 0.2, pre_unshuffled=True, grad_premasked=True, out_unshuffled=True)
        s1, u1 = B.ForkU2Fn16.apply(u1, 0.2)
        u2 = conv(u1, sn[2], None, 0.2, pre_unshuffled=True, grad_premasked=True, out_unshuffled=True)
        s2, u2 = B.ForkU2Fn16.apply(u2, 0.2)
        # conv3 / conv4 / conv5 feed only the resampling: its backward applies their LeakyReLU derivative (no stand-alone pass)
        x3 = conv(u2, sn[3], None, 0.2, pre_unshuffled=True, grad_premasked=True)
        x4 = conv(B.Bilinear2xFn16.apply(x3, None, 0.2), sn[4], None, 0.2, grad_premasked=True)
        x5 = conv(B.Bilinear2xFn16.apply(x4, s2 if skip else None, 0.2, True), sn[5], None, 0.2, grad_premasked=True)
        # conv6 adds the last skip in its epilogue (sign-keeping rounding: its LeakyReLU mask is recovered from the sum and x0)
        x6 = conv(B.Bilinear2xFn16.apply(x5, s1 if skip else None, 0.2, True), sn[6], None, 0.2, skip_u2=s0 if skip else None)
        out = conv(x6, sn[7], None, 0.2, grad_premasked=True)
        out = conv(out, sn[8], None, 0.2, input_slope=0.2, grad_premasked=True)
        return conv(out, self.conv9.weight, self.conv9.bias, 1.0, True, input_slope=0.2)  # fp32 NCHW logits

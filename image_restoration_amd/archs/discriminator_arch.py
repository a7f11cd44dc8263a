"""VGG-style discriminator on the HIP path.

Same constructor, forward contract and state_dict keys as the reference
``basicsr/archs/discriminator_arch.py:6-72`` (conv{i}_{0,1}, bn{i}_{0,1} incl. running statistics and
num_batches_tracked, linear1, linear2), so ``network_d: {type: VGGStyleDiscriminator128, ...}`` and saved
``net_d_*.pth`` files drop in.  The modules hold parameters only; the arithmetic is HIP launches: by default one whole-network
driver call per forward / backward (archs/vgg_disc_autograd.py -> sr_vgg_forward_* / sr_vgg_backward_*); the per-layer route
(hip_autograd.ConvFn / BNLReLUFn / LinearFn, ``forward_layers``) issues the same launches one autograd function at a time
and is kept as the cross-check of the drivers and for partially frozen networks.
"""
import ctypes as C
import math

import torch
from torch import nn
from torch.nn import init

from .. import _lib
from .. import hip_autograd as A
from ..utils.registry import ARCH_REGISTRY
from .arch_util import Conv3x3Params


class BatchNormParams(nn.Module):
    """State of nn.BatchNorm2d(c, affine=True): weight, bias, running_mean, running_var, num_batches_tracked
    (momentum 0.1, eps 1e-5 — the nn defaults the reference uses, discriminator_arch.py:23)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))

    def apply_lrelu(self, x, slope, bf16=False):
        """BatchNorm (batch statistics in train mode, running statistics in eval mode) + LeakyReLU on CB8 (or, bf16=True,
        on CB16 bf16 activations with fp32 parameters and statistics)."""
        if self.training:
            self.num_batches_tracked += 1
        if bf16:
            from ..hip_autograd_bf16 import BNLReLUFn16
            return BNLReLUFn16.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                     self.momentum, self.eps, slope)
        return A.BNLReLUFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                 self.momentum, self.eps, slope)

    def extra_repr(self):
        return f'{self.num_features}, eps={self.eps}, momentum={self.momentum} [HIP]'


class LinearParams(nn.Module):
    """weight [out, in] + bias [out] with nn.Linear's default initialisation."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        init.uniform_(self.bias, -bound, bound)

    def extra_repr(self):
        return f'in_features={self.in_features}, out_features={self.out_features} [HIP]'


@ARCH_REGISTRY.register()
class VGGStyleDiscriminator128(nn.Module):
    """VGGStyleDiscriminator128(num_in_ch, num_feat): [N, num_in_ch, 128, 128] -> [N, 1] logits."""

    input_size = 128  # VGGStyleDiscriminator256 adds one 8nf -> 8nf stage (discriminator_arch.py:75-143)

    def __init__(self, num_in_ch, num_feat, compute_dtype='fp32'):
        super().__init__()
        nf = num_feat
        self.num_in_ch, self.num_feat = num_in_ch, num_feat
        if compute_dtype not in ('fp32', 'bf16'):
            raise ValueError(f"compute dtype must be 'fp32' or 'bf16', got {compute_dtype!r}")
        # option key beyond the reference's: 'bf16' = CB16 bf16 activations / activation gradients on the generator's bf16
        # kernels (num_feat % 16 == 0); weights, BatchNorm statistics, weight gradients, linear layers and optimiser stay fp32
        assert compute_dtype == 'fp32' or nf % 16 == 0, 'bf16 needs num_feat to be a multiple of 16'
        self.compute_dtype = compute_dtype
        self.conv0_0 = Conv3x3Params(num_in_ch, nf, bias=True)
        self.conv0_1 = Conv3x3Params(nf, nf, bias=False, ksize=4)
        self.bn0_1 = BatchNormParams(nf)
        widths = [(nf, nf * 2), (nf * 2, nf * 4), (nf * 4, nf * 8), (nf * 8, nf * 8)] + ([(nf * 8, nf * 8)] if self.input_size == 256 else [])
        self.num_stages = len(widths)
        for i, (ci, co) in enumerate(widths, start=1):
            setattr(self, f'conv{i}_0', Conv3x3Params(ci, co, bias=False))
            setattr(self, f'bn{i}_0', BatchNormParams(co))
            setattr(self, f'conv{i}_1', Conv3x3Params(co, co, bias=False, ksize=4))
            setattr(self, f'bn{i}_1', BatchNormParams(co))
        self.linear1 = LinearParams(nf * 8 * 4 * 4, 100)
        self.linear2 = LinearParams(100, 1)
        self._packed = {}          # dtype -> (key, device blob of forward + data-gradient weight images)
        self._grad_sink = None     # set by optim.FlatAdam: gradients accumulate straight into its arena
        self._ws = None
        self._pack_epoch = 0

    # train-mode forwards are pure functions of (weights, input): models may run a repeated forward once (vgg_disc_autograd.py)
    repeatable_forward = True

    # ------------------------------------------------------------------ HIP plumbing (mirrors RRDBNet's)
    def invalidate_packed(self):
        """Parameter memory was rewritten behind autograd's back (fused Adam writes the arena through raw pointers)."""
        self._pack_epoch += 1

    def _cfg(self):
        return _lib.VGGCfg(self.num_in_ch, self.num_feat, self.input_size)

    def _param_list(self):
        cached = self.__dict__.get('_plist')
        if cached is not None and cached[0] is self.conv0_0.weight and cached[-1] is self.linear2.bias:
            return cached
        plist = [p for _, p in self.named_parameters()]
        self.__dict__['_plist'] = plist
        self.__dict__.pop('_bufptrs', None)
        return plist

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop('_plist', None)
        self.__dict__.pop('_bufptrs', None)
        return super()._apply(fn, *args, **kwargs)

    def _buffer_ptrs(self):
        """running_mean, running_var, num_batches_tracked of every BatchNorm in module order (sr_vgg_forward's host_buffers)."""
        bufs = [b for _, b in self.named_buffers()]
        key = tuple(b.data_ptr() for b in bufs)
        cached = self.__dict__.get('_bufptrs')
        if cached is None or cached[0] != key:
            lib = _lib.load()
            assert len(bufs) == 3 * lib.sr_vgg_num_batchnorm(C.byref(self._cfg())), len(bufs)
            for i, b in enumerate(bufs):
                assert b.dtype == (torch.int64 if i % 3 == 2 else torch.float32) and b.is_contiguous()
            cached = (key, (C.c_void_p * len(bufs))(*key))
            self.__dict__['_bufptrs'] = cached
        return cached[1]

    def _weights_key(self):
        """Identity of the current weights: parameter storage + torch version counters + the epochs of writers torch cannot
        see (FlatAdam's fused step, invalidate_packed, hip_ops.invalidate_packs)."""
        from .. import hip_ops
        params = self._param_list()
        return (self._pack_epoch, hip_ops._pack_epoch[0], getattr(params[0], '_sr_epoch', (0,))[0],
                tuple((p.data_ptr(), p._version) for p in params))

    def _ensure_packed(self, lib, cfg, stream, bf16):
        key = self._weights_key()
        hit = self._packed.get(bf16)
        if hit is not None and hit[0] == key:
            return hit[1]
        params = self._param_list()
        n = lib.sr_vgg_num_params(C.byref(cfg))
        if n != len(params):
            raise _lib.SrHipError(f'parameter count {len(params)} != {n} expected by libsr_hip.so')
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.SrHipError('discriminator parameters must be contiguous fp32 on one HIP device')
        nbytes = (lib.sr_vgg_packed_bytes_bf16 if bf16 else lib.sr_vgg_packed_bytes)(C.byref(cfg))
        blob = hit[1] if hit is not None and hit[1].numel() == nbytes and hit[1].device == dev else \
            torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ptrs = (C.c_void_p * n)(*[p.data_ptr() for p in params])
        _lib.check((lib.sr_vgg_pack_bf16 if bf16 else lib.sr_vgg_pack_f32)(C.byref(cfg), ptrs, blob.data_ptr(), stream), 'sr_vgg_pack')
        self._packed[bf16] = (key, blob)
        return blob

    def _workspace(self, lib, cfg, n, dev, bf16):
        nbytes = (lib.sr_vgg_workspace_bytes_bf16 if bf16 else lib.sr_vgg_workspace_bytes)(C.byref(cfg), n)
        ws = self._ws
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            ws = self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return ws, nbytes

    def _check_input(self, x):
        sz = self.input_size
        assert x.size(2) == sz and x.size(3) == sz, (f'Input spatial size must be {sz}x{sz}, but received {x.size()}.')
        if not x.is_cuda:
            raise _lib.SrHipError(f'{type(self).__name__}.forward runs only on a HIP device (no CPU fallback)')

    def forward(self, x, kept=None, slot=None):
        """``kept`` / ``slot``: see archs/vgg_disc_autograd.vgg_apply (a model that knows two calls see the same input and
        weights hands the first call's KeptForward to the second)."""
        self._check_input(x)
        flags = {p.requires_grad for p in self._param_list()}
        if len(flags) > 1 and torch.is_grad_enabled():
            return self.forward_layers(x)   # partially frozen: the per-layer route handles any mixture
        from .vgg_disc_autograd import vgg_apply
        return vgg_apply(self, x, kept, slot)

    def forward_layers(self, x):
        """The same network one autograd function per layer (the route of rounds 1-3): same launches, same bits."""
        self._check_input(x)
        if self.compute_dtype == 'bf16':
            return self._forward_bf16(x)
        feat = A.ToCB8.apply(x.contiguous().float())
        feat = A.ConvFn.apply(feat, self.conv0_0.weight, self.conv0_0.bias, 0.2)            # lrelu(conv0_0(x))
        feat = self.bn0_1.apply_lrelu(A.ConvFn.apply(feat, self.conv0_1.weight, None, 1.0), 0.2)  # 64x64
        for i in range(1, self.num_stages + 1):
            c0, b0 = getattr(self, f'conv{i}_0'), getattr(self, f'bn{i}_0')
            c1, b1 = getattr(self, f'conv{i}_1'), getattr(self, f'bn{i}_1')
            feat = b0.apply_lrelu(A.ConvFn.apply(feat, c0.weight, None, 1.0), 0.2)
            feat = b1.apply_lrelu(A.ConvFn.apply(feat, c1.weight, None, 1.0), 0.2)           # 32, 16, 8, 4
        feat = A.FromCB8.apply(feat, self.num_feat * 8)
        feat = feat.reshape(feat.size(0), -1)                                                # view(N, -1): plain metadata
        feat = A.LinearFn.apply(feat, self.linear1.weight, self.linear1.bias, 0.2)
        return A.LinearFn.apply(feat, self.linear2.weight, self.linear2.bias, 1.0)

    def _forward_bf16(self, x):
        from .. import hip_autograd_bf16 as B

        def conv(t, p, slope):
            return B.ConvFn16.apply(t, p.weight, p.bias, slope, False)
        feat = conv(B.ToCB16.apply(x.contiguous().float()), self.conv0_0, 0.2)                # lrelu(conv0_0(x))
        feat = self.bn0_1.apply_lrelu(conv(feat, self.conv0_1, 1.0), 0.2, bf16=True)           # 64x64
        for i in range(1, self.num_stages + 1):
            feat = getattr(self, f'bn{i}_0').apply_lrelu(conv(feat, getattr(self, f'conv{i}_0'), 1.0), 0.2, bf16=True)
            feat = getattr(self, f'bn{i}_1').apply_lrelu(conv(feat, getattr(self, f'conv{i}_1'), 1.0), 0.2, bf16=True)
        feat = B.FromCB16.apply(feat, self.num_feat * 8)                                        # fp32 NCHW for the linear head
        feat = feat.reshape(feat.size(0), -1)
        feat = A.LinearFn.apply(feat, self.linear1.weight, self.linear1.bias, 0.2)
        return A.LinearFn.apply(feat, self.linear2.weight, self.linear2.bias, 1.0)


@ARCH_REGISTRY.register()
class VGGStyleDiscriminator256(VGGStyleDiscriminator128):
    """VGGStyleDiscriminator256(num_in_ch, num_feat): [N, num_in_ch, 256, 256] -> [N, 1] logits; the 128 network with a sixth
    stage conv5_0 / bn5_0 / conv5_1 / bn5_1 (8nf -> 8nf) before the same linear head (discriminator_arch.py:75-143)."""

    input_size = 256


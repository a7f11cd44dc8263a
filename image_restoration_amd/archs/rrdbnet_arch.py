"""RRDBNet (ESRGAN generator) on the MI355X HIP path.

Same constructor, forward contract and state_dict keys as the reference
``basicsr/archs/rrdbnet_arch.py`` (ResidualDenseBlock :9-39, RRDB :42-63, RRDBNet :66-119),
so ``network_g: {type: RRDBNet, ...}`` option blocks and ESRGAN checkpoints drop in.  The
modules below only hold parameters; ``RRDBNet.forward`` hands the whole network to
``sr_rrdbnet_forward_f32`` (include/sr_hip.h), which runs it as fused MFMA conv launches on
the current HIP stream.
"""
import ctypes as C
import os

import torch
from torch import nn

from .. import _lib
from ..utils.registry import ARCH_REGISTRY
from .arch_util import Conv3x3Params, default_init_weights, make_layer


class ResidualDenseBlock(nn.Module):
    """Parameters of one residual dense block: conv1..conv5 with growing input width
    (reference :19-30); RDB convs are initialised kaiming_normal * 0.1, bias 0."""

    def __init__(self, num_feat=64, num_grow_ch=32):
        super().__init__()
        self.conv1 = Conv3x3Params(num_feat, num_grow_ch)
        self.conv2 = Conv3x3Params(num_feat + num_grow_ch, num_grow_ch)
        self.conv3 = Conv3x3Params(num_feat + 2 * num_grow_ch, num_grow_ch)
        self.conv4 = Conv3x3Params(num_feat + 3 * num_grow_ch, num_grow_ch)
        self.conv5 = Conv3x3Params(num_feat + 4 * num_grow_ch, num_feat)
        default_init_weights([self.conv1, self.conv2, self.conv3, self.conv4, self.conv5], 0.1)


class RRDB(nn.Module):
    """Three residual dense blocks (reference :52-56)."""

    def __init__(self, num_feat, num_grow_ch=32):
        super().__init__()
        self.rdb1 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb2 = ResidualDenseBlock(num_feat, num_grow_ch)
        self.rdb3 = ResidualDenseBlock(num_feat, num_grow_ch)


_DEBUG_PARAM_LIST = os.environ.get('SR_DEBUG_PACKS') == '1'


@ARCH_REGISTRY.register()
class RRDBNet(nn.Module):
    """RRDBNet(num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32[, compute_dtype='fp32']).

    forward(x[N, num_in_ch, H, W] fp32 on a HIP device) -> [N, num_out_ch, 4H/s', 4W/s']
    with s' = 1, 2, 4 for scale 4, 2, 1 (pixel_unshuffle at the input, reference :90-93,106-109).
    """

    def __init__(self, num_in_ch, num_out_ch, scale=4, num_feat=64, num_block=23, num_grow_ch=32, compute_dtype='fp32'):
        super().__init__()
        self.scale = scale
        self.num_in_ch, self.num_out_ch = num_in_ch, num_out_ch
        self.num_feat, self.num_block, self.num_grow_ch = num_feat, num_block, num_grow_ch
        if scale == 2:
            num_in_ch = num_in_ch * 4
        elif scale == 1:
            num_in_ch = num_in_ch * 16
        self.conv_first = Conv3x3Params(num_in_ch, num_feat)
        self.body = make_layer(RRDB, num_block, num_feat=num_feat, num_grow_ch=num_grow_ch)
        self.conv_body = Conv3x3Params(num_feat, num_feat)
        self.conv_up1 = Conv3x3Params(num_feat, num_feat)
        self.conv_up2 = Conv3x3Params(num_feat, num_feat)
        self.conv_hr = Conv3x3Params(num_feat, num_feat)
        self.conv_last = Conv3x3Params(num_feat, num_out_ch)
        self._packed = None       # device blob of MFMA-ready weights
        self._packed_key = None   # (data_ptr, _version) of every parameter at pack time
        self._packed_dg = None
        self._packed_dg_key = None
        self._grad_sink = None    # set by optim.FlatAdam: gradients accumulate straight into its arena
        self._workspaces = {}
        self.compute_dtype = 'fp32'
        self._packed_h = None
        self._packed_h_key = None
        self.set_compute_dtype(compute_dtype)  # option key beyond the reference's: 'bf16' = reduced-precision kernels

    # ------------------------------------------------------------------ HIP plumbing
    def invalidate_packed(self):
        """Call after parameter memory was updated behind autograd's back (fused Adam / EMA kernels write the
        arena without bumping tensor versions)."""
        self._packed_key = None
        self._packed_dg_key = None
        self._packed_h_key = None

    def set_compute_dtype(self, dtype):
        """'fp32' (reference numerics, default) or 'bf16' (bf16 activations, activation gradients and weight images on
        v_mfma_f32_32x32x16_bf16; fp32 master weights, accumulation, parameter gradients and optimiser)."""
        if dtype not in ('fp32', 'bf16'):
            raise ValueError(f"compute dtype must be 'fp32' or 'bf16', got {dtype!r}")
        self.compute_dtype = dtype
        return self

    def _cfg(self):
        # scale other than 1/2/4 behaves like 4 in the reference (no unshuffle, :106-111)
        s = self.scale if self.scale in (1, 2) else 4
        return _lib.RRDBNetCfg(self.num_in_ch, self.num_out_ch, s, self.num_feat, self.num_block, self.num_grow_ch)

    def _param_list(self):
        """Parameters in state_dict order (what sr_rrdbnet_pack_f32 expects).  The walk over the module tree (702 parameters, twice
        per training step: 2.6 ms of host time) is cached; the cache is dropped when the first or the last parameter object is no
        longer the module's (``load_state_dict(assign=True)``, a re-registered parameter) and by ``_apply`` (``.to()``, ``.cuda()``)."""
        cached = self.__dict__.get('_plist')
        if cached is not None and cached[0] is self.conv_first.weight and cached[-1] is self.conv_last.bias:
            if _DEBUG_PARAM_LIST:   # SR_DEBUG_PACKS=1: the full walk every time, and say so if the shortcut would have lied
                fresh = [p for _, p in self.named_parameters()]
                assert len(fresh) == len(cached) and all(a is b for a, b in zip(fresh, cached)), \
                    'a parameter in the middle of the network was re-registered: call net._apply(lambda t: t) or invalidate the list'
            return cached
        plist = [p for _, p in self.named_parameters()]
        self.__dict__['_plist'] = plist
        return plist

    def _apply(self, fn, *args, **kwargs):
        self.__dict__.pop('_plist', None)
        return super()._apply(fn, *args, **kwargs)

    def _ensure_packed(self, lib, cfg, stream):
        params = self._param_list()
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._packed is not None and key == self._packed_key:
            return self._packed
        n = lib.sr_rrdbnet_num_params(C.byref(cfg))
        if n != len(params):
            raise _lib.SrHipError(f'parameter count {len(params)} != {n} expected by libsr_hip.so')
        dev = params[0].device
        for p in params:
            if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.SrHipError('RRDBNet parameters must be contiguous fp32 on one HIP device')
        nbytes = lib.sr_rrdbnet_packed_bytes(C.byref(cfg))
        if self._packed is None or self._packed.numel() != nbytes or self._packed.device != dev:
            self._packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ptrs = (C.c_void_p * n)(*[p.data_ptr() for p in params])
        _lib.check(lib.sr_rrdbnet_pack_f32(C.byref(cfg), ptrs, self._packed.data_ptr(), stream), 'sr_rrdbnet_pack_f32')
        self._packed_key = key
        return self._packed

    def _ensure_packed_dgrad(self, lib, cfg, stream, bf16=False):
        """Transposed/flipped weight images for the data-gradient convs, cached per parameter version (and dtype)."""
        params = self._param_list()
        key = (bf16,) + tuple((p.data_ptr(), p._version) for p in params[0::2])
        if getattr(self, '_packed_dg', None) is not None and key == self._packed_dg_key:
            return self._packed_dg
        dev = params[0].device
        nbytes = (lib.sr_rrdbnet_packed_dgrad_bytes_bf16 if bf16 else lib.sr_rrdbnet_packed_dgrad_bytes)(C.byref(cfg))
        if getattr(self, '_packed_dg', None) is None or self._packed_dg.numel() != nbytes or self._packed_dg.device != dev:
            self._packed_dg = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ptrs = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        pack = lib.sr_rrdbnet_pack_dgrad_bf16 if bf16 else lib.sr_rrdbnet_pack_dgrad_f32
        _lib.check(pack(C.byref(cfg), ptrs, self._packed_dg.data_ptr(), stream), 'sr_rrdbnet_pack_dgrad')
        self._packed_dg_key = key
        return self._packed_dg

    def _bwd_workspace(self, nbytes, dev):
        ws = getattr(self, '_bwd_ws', None)
        if ws is None or ws.numel() < nbytes or ws.device != dev:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._bwd_ws = ws
        return ws

    def _workspace(self, lib, cfg, n, h, w, dev):
        nbytes = lib.sr_rrdbnet_workspace_bytes(C.byref(cfg), n, h, w)
        if nbytes == 0:
            u = {4: 1, 2: 2, 1: 4}[cfg.scale]
            # the reference asserts divisibility inside pixel_unshuffle (arch_util.py:197)
            assert h % u == 0 and w % u == 0, f'input {h}x{w} is not divisible by the pixel_unshuffle factor {u}'
            raise _lib.SrHipError('sr_rrdbnet_workspace_bytes returned 0')
        key = (n, h, w, str(dev))
        ws = self._workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            self._workspaces.clear()  # one live shape at a time keeps HBM use bounded
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._workspaces[key] = ws
        return ws, nbytes

    def forward(self, x):
        if not x.is_cuda:
            raise _lib.SrHipError('RRDBNet.forward runs only on a HIP device (no CPU fallback): move the module '
                                  'and input with .to("cuda")')
        if x.dim() != 4 or x.size(1) != self.num_in_ch:
            raise ValueError(f'expected [N, {self.num_in_ch}, H, W], got {tuple(x.shape)}')
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._param_list())):
            from .rrdbnet_autograd import rrdbnet_apply
            return rrdbnet_apply(self, x)
        return self._forward_inference(x)

    def _forward_inference(self, x):
        lib = _lib.load()
        x = x.contiguous().float()
        n, _, h, w = x.shape
        cfg = self._cfg()
        with torch.cuda.device(x.device):
            stream = torch.cuda.current_stream().cuda_stream
            return self._launch(lib, cfg, x, n, h, w, stream)

    def _ensure_packed_bf16(self, lib, cfg, stream):
        """bf16 MFMA weight images, rounded from the fp32 master parameters; cached per parameter version."""
        params = self._param_list()
        key = tuple((p.data_ptr(), p._version) for p in params)
        if self._packed_h is None or key != self._packed_h_key:
            dev = params[0].device
            nb = lib.sr_rrdbnet_packed_bytes_bf16(C.byref(cfg))
            if self._packed_h is None or self._packed_h.numel() != nb or self._packed_h.device != dev:
                self._packed_h = torch.empty(nb, dtype=torch.uint8, device=dev)
            for p in params:
                if p.device != dev or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.SrHipError('RRDBNet parameters must be contiguous fp32 on one HIP device')
            ptrs = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
            _lib.check(lib.sr_rrdbnet_pack_bf16(C.byref(cfg), ptrs, self._packed_h.data_ptr(), stream),
                       'sr_rrdbnet_pack_bf16')
            self._packed_h_key = key
        return self._packed_h

    def _launch_bf16(self, lib, cfg, x, n, h, w, stream):
        self._ensure_packed_bf16(lib, cfg, stream)
        nbytes = lib.sr_rrdbnet_workspace_bytes_bf16(C.byref(cfg), n, h, w)
        if nbytes == 0:
            raise _lib.SrHipError(f'sr_rrdbnet_workspace_bytes_bf16 returned 0 for input {h}x{w}')
        wkey = ('bf16', n, h, w, str(x.device))
        ws = self._workspaces.get(wkey)
        if ws is None:
            self._workspaces.clear()
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            self._workspaces[wkey] = ws
        up = {4: 4, 2: 2, 1: 1}[cfg.scale]
        y = torch.empty((n, self.num_out_ch, h * up, w * up), dtype=torch.float32, device=x.device)
        _lib.check(
            lib.sr_rrdbnet_forward_bf16(C.byref(cfg), self._packed_h.data_ptr(), x.data_ptr(), y.data_ptr(), n, h, w,
                                        ws.data_ptr(), nbytes, stream), 'sr_rrdbnet_forward_bf16')
        return y

    def _launch(self, lib, cfg, x, n, h, w, stream):
        if self.compute_dtype == 'bf16':
            return self._launch_bf16(lib, cfg, x, n, h, w, stream)
        packed = self._ensure_packed(lib, cfg, stream)
        ws, nbytes = self._workspace(lib, cfg, n, h, w, x.device)
        up = {4: 4, 2: 2, 1: 1}[cfg.scale]
        y = torch.empty((n, self.num_out_ch, h * up, w * up), dtype=torch.float32, device=x.device)
        _lib.check(
            lib.sr_rrdbnet_forward_f32(C.byref(cfg), packed.data_ptr(), x.data_ptr(), y.data_ptr(), n, h, w,
                                       ws.data_ptr(), nbytes, stream), 'sr_rrdbnet_forward_f32')
        return y

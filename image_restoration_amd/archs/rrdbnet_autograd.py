"""Autograd bridge of the HIP generator: ONE torch.autograd.Function for the whole RRDBNet.

forward  -> sr_rrdbnet_forward_train_f32 (keeps the activations in a device workspace)
backward -> sr_rrdbnet_backward_f32 (data + weight gradients of all 351 convs as HIP launches)
With ``net.set_compute_dtype('bf16')`` the *_bf16 twins run instead (bf16 activations and activation gradients,
fp32 master weights, fp32 parameter gradients); x, y and every tensor autograd sees stay fp32.

This is what stands where the reference relies on autograd through nn.Conv2d / LeakyReLU / cat /
interpolate (rrdbnet_arch.py:105-119 under esrgan_model.py:18,47).  Parameter gradients are
returned to autograd as ordinary tensors, so DistributedDataParallel's reducer, optimizers and
``requires_grad_(False)`` toggling (esrgan_model.py:14-15) work unchanged.
"""
import ctypes as C

import torch

from .. import _lib


class _RRDBNetFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, net, x, *params):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.SrHipError('RRDBNet runs only on a HIP device (no CPU fallback)')
        x = x.contiguous().float()
        n, _, h, w = x.shape
        cfg = net._cfg()
        dev = x.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            bf16 = net.compute_dtype == 'bf16'
            packed = net._ensure_packed_bf16(lib, cfg, stream) if bf16 else net._ensure_packed(lib, cfg, stream)
            nbytes = (lib.sr_rrdbnet_saved_bytes_bf16 if bf16 else lib.sr_rrdbnet_saved_bytes)(C.byref(cfg), n, h, w)
            if nbytes == 0:
                u = {4: 1, 2: 2, 1: 4}[cfg.scale]
                assert h % u == 0 and w % u == 0, f'input {h}x{w} is not divisible by the pixel_unshuffle factor {u}'
                raise _lib.SrHipError('sr_rrdbnet_saved_bytes returned 0')
            saved = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            up = {4: 4, 2: 2, 1: 1}[cfg.scale]
            y = torch.empty((n, net.num_out_ch, h * up, w * up), dtype=torch.float32, device=dev)
            fwd = lib.sr_rrdbnet_forward_train_bf16 if bf16 else lib.sr_rrdbnet_forward_train_f32
            _lib.check(fwd(C.byref(cfg), packed.data_ptr(), x.data_ptr(), y.data_ptr(), n, h, w, saved.data_ptr(), nbytes,
                           stream), 'sr_rrdbnet_forward_train_' + ('bf16' if bf16 else 'f32'))
        ctx.net, ctx.cfg, ctx.saved, ctx.shape, ctx.bf16 = net, cfg, saved, (n, h, w), bf16
        ctx.x_shape = tuple(x.shape)
        ctx.param_versions = tuple(p._version for p in params)
        ctx.params = params
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        net, cfg, (n, h, w), bf16 = ctx.net, ctx.cfg, ctx.shape, ctx.bf16
        params = ctx.params
        dy = dy.contiguous().float()
        dev = dy.device
        need_x = ctx.needs_input_grad[1]
        need_p = ctx.needs_input_grad[2:]
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            # data-gradient weight images (transposed / flipped), cached per parameter version
            packed_dg = net._ensure_packed_dgrad(lib, cfg, stream, bf16)
            # deferred weight gradients (NetPack.defer_weight_gradients): the call returns with the lane still busy; whoever consumes
            # the gradient arena joins first (NetPack.update), and what the lane reads stays alive until then
            defer = bool(bf16 and getattr(net, '_defer_wgrad', False) and getattr(net, '_grad_sink', None) is not None)
            if bf16:
                lib.sr_set_backward_wgrad_deferred(int(getattr(net, '_defer_mode', 1)) if defer else 0)
            wbytes = (lib.sr_rrdbnet_backward_workspace_bytes_bf16 if bf16 else
                      lib.sr_rrdbnet_backward_workspace_bytes)(C.byref(cfg), n, h, w)
            ws = net._bwd_workspace(wbytes, dev)
            sink = getattr(net, '_grad_sink', None)
            if sink is not None and any(need_p):
                # Flat-arena mode (image_restoration_amd.optim.FlatAdam): gradients are ACCUMULATED straight into
                # the arena that is all-reduced and consumed by the fused Adam kernel; autograd sees no grads.
                if not all(need_p):
                    raise _lib.SrHipError('flat-arena mode needs every generator parameter to require grad')
                grads = [None] * len(params)
                ptrs = (C.c_void_p * len(params))(*sink.grad_ptrs)
                accumulate = 1
            else:
                grads = [torch.empty_like(p) if need else None for p, need in zip(params, need_p)]
                # weight and bias of one conv travel together: a conv is skipped only when its weight needs no grad
                ptrs = (C.c_void_p * len(params))(*[g.data_ptr() if g is not None else None for g in grads])
                for i in range(0, len(params), 2):
                    if grads[i] is None and grads[i + 1] is not None:
                        raise _lib.SrHipError('bias.requires_grad without weight.requires_grad is not supported')
                accumulate = 0
            dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dev) if need_x else None
            bwd = lib.sr_rrdbnet_backward_bf16 if bf16 else lib.sr_rrdbnet_backward_f32
            try:
                _lib.check(bwd(C.byref(cfg), packed_dg.data_ptr(), ctx.saved.data_ptr(), ctx.saved.numel(), dy.data_ptr(), n, h, w,
                               ptrs, dx.data_ptr() if dx is not None else None, ws.data_ptr(), wbytes, accumulate, stream),
                           'sr_rrdbnet_backward_' + ('bf16' if bf16 else 'f32'))
            finally:
                if defer:
                    lib.sr_set_backward_wgrad_deferred(0)
            if defer:
                net._lane_holds.append((ctx.saved, ws, packed_dg))   # released by NetPack.update after the join
        ctx.saved = None
        return (None, dx) + tuple(grads)


def rrdbnet_apply(net, x):
    return _RRDBNetFunction.apply(net, x, *net._param_list())

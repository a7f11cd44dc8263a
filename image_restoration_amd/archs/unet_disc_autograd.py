"""Autograd bridge of the bf16 U-Net discriminator: ONE torch.autograd.Function for the convolutional network.

forward  -> sr_unet_pack_bf16 + sr_unet_forward_bf16   (this forward's effective weights packed once, activations kept in one block)
backward -> sr_unet_backward_bf16                      (data and weight gradients of all ten convs)

Spectral normalisation stays a Function of its own in front of this one (hip_autograd.SpectralNormBatchFn): the tensors handed in
here are the normalised weights of THIS forward, and the gradients handed back are the gradients wrt them.  Same launches, same order
per tensor as the per-layer route (UNetDiscriminatorSN.forward_layers): bit-identical; what changes is one C call per pass instead of
~25 / ~45 Python autograd applies."""
import ctypes as C

import torch

from .. import _lib


class _UNetFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, net, x, *weights):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.SrHipError('UNetDiscriminatorSN runs only on a HIP device (no CPU fallback)')
        x = x.contiguous().float()
        n, _, h, w = x.shape
        cfg = net._cfg()
        dev = x.device
        ws = [t.detach().contiguous().float() for t in weights]
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            packed = torch.empty(lib.sr_unet_packed_bytes_bf16(C.byref(cfg)), dtype=torch.uint8, device=dev)
            ptrs = (C.c_void_p * len(ws))(*[t.data_ptr() for t in ws])
            _lib.check(lib.sr_unet_pack_bf16(C.byref(cfg), ptrs, packed.data_ptr(), stream), 'sr_unet_pack_bf16')
            nbytes = lib.sr_unet_saved_bytes_bf16(C.byref(cfg), n, h, w)
            if nbytes == 0:
                raise _lib.SrHipError(f'sr_unet_saved_bytes_bf16 returned 0 for input {h}x{w}')
            saved = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            logits = torch.empty((n, 1, h, w), dtype=torch.float32, device=dev)
            _lib.check(lib.sr_unet_forward_bf16(C.byref(cfg), packed.data_ptr(), x.data_ptr(), logits.data_ptr(), n, h, w, saved.data_ptr(),
                                                nbytes, stream), 'sr_unet_forward_bf16')
        ctx.net, ctx.cfg, ctx.packed, ctx.saved, ctx.shape = net, cfg, packed, saved, (n, h, w)
        ctx.x_shape = tuple(x.shape)
        ctx.w_shapes = [tuple(t.shape) for t in weights]
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        net, cfg, (n, h, w) = ctx.net, ctx.cfg, ctx.shape
        dlogits = dlogits.contiguous().float()
        dev = dlogits.device
        need_x = ctx.needs_input_grad[1]
        need_w = ctx.needs_input_grad[2:]
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            wbytes = lib.sr_unet_workspace_bytes_bf16(C.byref(cfg), n, h, w)
            ws = net._workspace(wbytes, dev)
            grads = [None] * len(ctx.w_shapes)
            dptrs = None
            if any(need_w):
                if not all(need_w):
                    raise _lib.SrHipError('the whole-network U-Net backward needs all weights to require grad or none')
                grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.w_shapes]
                dptrs = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
            dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dev) if need_x else None
            _lib.check(lib.sr_unet_backward_bf16(C.byref(cfg), ctx.packed.data_ptr(), ctx.saved.data_ptr(), ctx.saved.numel(),
                                                 dlogits.data_ptr(), n, h, w, dptrs, dx.data_ptr() if dx is not None else None, ws.data_ptr(),
                                                 wbytes, stream), 'sr_unet_backward_bf16')
        ctx.saved = ctx.packed = None
        return (None, dx) + tuple(grads)


def unet_apply(net, x, weights):
    return _UNetFunction.apply(net, x, *weights)

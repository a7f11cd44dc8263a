"""Autograd bridge of the HIP VGG-style discriminator: ONE torch.autograd.Function for the whole network.

forward  -> sr_vgg_forward_{f32,bf16}   (keeps the activations the backward needs in one device block)
backward -> sr_vgg_backward_{f32,bf16}  (data and weight gradients of all layers; parameter gradients go straight into
                                         the FlatAdam arena when there is one, like the generator's)

This stands where the reference relies on autograd through nn.Conv2d / BatchNorm2d / LeakyReLU / Linear
(discriminator_arch.py:52-72 under esrgan_model.py:38-47,65-72).  What the block buys beyond fewer host calls: a
``KeptForward`` can be handed to a later call on the SAME input and the SAME weights — the launches are deterministic and
train-mode BatchNorm does not read its running statistics, so the repeat is bit-identical; the call then only replays the
running-statistics update the reference's forward would have made (sr_vgg_apply_stats_*) and shares the activations with
whatever backward follows.  models/srgan_model.py decides when that is the case.
"""
import ctypes as C

import torch

from .. import _lib


class KeptForward:
    """The device state one train-mode forward left behind: activations + BatchNorm batch statistics (``saved``), the
    logits, and what identifies the call (input storage / version, weight epoch) so that a stale hand-over is refused."""

    __slots__ = ('saved', 'logits', 'n', 'bf16', 'train', 'x_key', 'w_key')

    def matches(self, x, w_key, bf16, train):
        return (self.x_key == _x_key(x) and self.w_key == w_key and self.bf16 == bf16 and self.train == train)


def _x_key(x):
    return (x.data_ptr(), x._version, tuple(x.shape), str(x.device))


def _fn(lib, name, bf16):
    return getattr(lib, name + ('_bf16' if bf16 else '_f32'))


class _VGGFunction(torch.autograd.Function):

    @staticmethod
    def forward(ctx, net, x, kept, slot, *params):
        lib = _lib.load()
        if not x.is_cuda:
            raise _lib.SrHipError(f'{type(net).__name__} runs only on a HIP device (no CPU fallback)')
        x = x.contiguous().float()
        n = x.size(0)
        cfg = net._cfg()
        bf16 = net.compute_dtype == 'bf16'
        train = bool(net.training)
        dev = x.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            bufs = net._buffer_ptrs()
            if kept is not None:
                # a repeat of a forward that already ran on this input with these weights: only the statistics move
                if train:
                    _lib.check(_fn(lib, 'sr_vgg_apply_stats', bf16)(C.byref(cfg), kept.saved.data_ptr(), kept.saved.numel(), n, bufs, 1,
                                                                    stream), 'sr_vgg_apply_stats')
            else:
                packed = net._ensure_packed(lib, cfg, stream, bf16)
                nbytes = (lib.sr_vgg_saved_bytes_bf16 if bf16 else lib.sr_vgg_saved_bytes)(C.byref(cfg), n)
                if nbytes == 0:
                    raise _lib.SrHipError('sr_vgg_saved_bytes returned 0')
                kept = KeptForward()
                kept.saved = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                kept.logits = torch.empty((n, 1), dtype=torch.float32, device=dev)
                kept.n, kept.bf16, kept.train = n, bf16, train
                kept.x_key, kept.w_key = _x_key(x), net._weights_key()
                ws, wbytes = net._workspace(lib, cfg, n, dev, bf16)
                pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
                _lib.check(_fn(lib, 'sr_vgg_forward', bf16)(C.byref(cfg), packed.data_ptr(), pp, bufs, x.data_ptr(), kept.logits.data_ptr(),
                                                            n, int(train), kept.saved.data_ptr(), nbytes, ws.data_ptr(), wbytes, stream),
                           'sr_vgg_forward')
            if slot is not None:
                slot.append(kept)
        ctx.net, ctx.cfg, ctx.kept, ctx.params, ctx.x_shape = net, cfg, kept, params, tuple(x.shape)
        return kept.logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        net, cfg, kept, params = ctx.net, ctx.cfg, ctx.kept, ctx.params
        bf16, n = kept.bf16, kept.n
        dlogits = dlogits.contiguous().float()
        dev = dlogits.device
        need_x = ctx.needs_input_grad[1]
        need_p = ctx.needs_input_grad[4:]
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream().cuda_stream
            if kept.w_key != net._weights_key():
                raise _lib.SrHipError('the discriminator weights changed between a forward and its backward')
            packed = net._ensure_packed(lib, cfg, stream, bf16)
            ws, wbytes = net._workspace(lib, cfg, n, dev, bf16)
            sink = getattr(net, '_grad_sink', None)
            grads = [None] * len(params)
            dptrs = None
            accumulate = 0
            if any(need_p):
                if not all(need_p):
                    raise _lib.SrHipError('the whole-network discriminator backward needs all parameters to require grad or none')
                if sink is not None:   # flat-arena mode: accumulate into the arena FlatAdam all-reduces and consumes
                    dptrs = (C.c_void_p * len(params))(*sink.grad_ptrs)
                    accumulate = 1
                else:
                    grads = [torch.empty_like(p) for p in params]
                    dptrs = (C.c_void_p * len(params))(*[g.data_ptr() for g in grads])
            dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dev) if need_x else None
            pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
            _lib.check(_fn(lib, 'sr_vgg_backward', bf16)(C.byref(cfg), packed.data_ptr(), pp, kept.saved.data_ptr(), kept.saved.numel(),
                                                         dlogits.data_ptr(), n, int(kept.train), dptrs, accumulate,
                                                         dx.data_ptr() if dx is not None else None, ws.data_ptr(), wbytes, stream),
                       'sr_vgg_backward')
        return (None, dx, None, None) + tuple(grads)


def vgg_apply(net, x, kept=None, slot=None):
    """logits = net(x) through the whole-network drivers.  ``kept``: a KeptForward of the same input and weights (the caller's
    claim; checked) — the forward is not run again.  ``slot``: a list that receives this call's KeptForward."""
    if kept is not None and not kept.matches(x, net._weights_key(), net.compute_dtype == 'bf16', bool(net.training)):
        kept = None
    return _VGGFunction.apply(net, x, kept, slot, *net._param_list())

"""torch.autograd.Function wrappers of the single-op C ABI (discriminators, losses).

Activations travel between these Functions as plain fp32 tensors shaped [N, C/8, H, W, 8]
(the CB8 layout of include/sr_hip.h); every forward/backward below is libsr_hip.so launches
only.  The generator does not use this file: it is one fused Function
(archs/rrdbnet_autograd.py).
"""
import ctypes as C

import torch

from . import _lib
from . import hip_ops as H

scratch = H.scratch


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _cb8(t):
    return H.CB8(t)


class ToCB8(torch.autograd.Function):
    """NCHW -> CB8 (sr_nchw_to_cb8_f32); backward CB8 -> NCHW."""

    @staticmethod
    def forward(ctx, x):
        ctx.c = x.size(1)
        return H.nchw_to_cb8(x).buf

    @staticmethod
    def backward(ctx, g):
        return H.cb8_to_nchw(_cb8(g.contiguous()), ctx.c)


class FromCB8(torch.autograd.Function):
    """CB8 -> NCHW with `channels` real channels; backward NCHW -> CB8 (pad channels zero)."""

    @staticmethod
    def forward(ctx, t, channels):
        ctx.cb = t.size(1)
        return H.cb8_to_nchw(_cb8(t), channels)

    @staticmethod
    def backward(ctx, g):
        out = H.CB8.empty(g.size(0), ctx.cb * 8, g.size(2), g.size(3), g.device)
        return H.nchw_to_cb8(g.contiguous(), out=out).buf, None


class ConvFn(torch.autograd.Function):
    """3x3/s1/p1 or 4x4/s2/p1 convolution (+bias, +LeakyReLU(act_slope)) on CB8.

    forward  : sr_conv3x3_f32 / sr_conv4x4s2_f32
    backward : LeakyReLU mask + data gradient fused in sr_conv3x3_f32 (mode-1 weights) / sr_conv4x4s2_dgrad_f32,
               weight gradient sr_conv3x3_wgrad_f32 / sr_conv4x4s2_wgrad_f32.
    """

    @staticmethod
    def forward(ctx, x, weight, bias, act_slope):
        k = weight.size(2)
        src = _cb8(x)
        if k == 3:
            out = H.conv3x3(src, H.cached_pack('f32 fwd', weight, bias, lambda: H.PackedConv(weight, bias)), act_slope=act_slope)
        elif k == 4:
            out = H.conv4x4s2(src, H.cached_pack('f32 fwd', weight, bias, lambda: H.PackedConv4x4s2(weight, bias)), act_slope=act_slope)
        else:
            raise NotImplementedError(f'kernel size {k}')
        ctx.save_for_backward(x, weight, out.buf if act_slope != 1.0 else None)
        ctx.act_slope, ctx.has_bias, ctx.k = act_slope, bias is not None, k
        ctx.param = weight if isinstance(weight, torch.nn.Parameter) else None   # key of the transposed image's cache entry
        return out.buf

    @staticmethod
    def backward(ctx, gy):
        x, weight, y = ctx.saved_tensors
        k = ctx.k
        cout, cin = weight.shape[:2]
        gy = gy.contiguous()
        dev = gy.device
        # dL/d(pre-activation): LeakyReLU backward against the saved output
        if y is not None:
            lib = _lib.load()
            dz = torch.empty_like(gy)
            with torch.cuda.device(dev):
                _lib.check(lib.sr_lrelu_bwd_f32(gy.data_ptr(), y.data_ptr(), dz.data_ptr(), ctx.act_slope, gy.numel(),
                                                _stream(dev)), 'sr_lrelu_bwd_f32')
        else:
            dz = gy
        dzc, src = _cb8(dz), _cb8(x)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dx = dw = db = None
        if need_x:
            if k == 3:
                dx = H.conv3x3(dzc, H.cached_pack('f32 dgrad', ctx.param, None, lambda: H.PackedConv(weight, None, mode=1))).buf
            else:
                dx = H.conv4x4s2_dgrad(dzc, H.cached_pack('f32 dgrad', ctx.param, None, lambda: H.PackedConv4x4s2(weight, None, mode=1)),
                                       src.h, src.w).buf
            if dx.size(1) != x.size(1):  # dgrad writes roundup8(cin) channels == x's blocks
                dx = dx[:, :x.size(1)].contiguous()
        if need_w or (need_b and ctx.has_bias):
            if k == 3:
                dw, db = H.conv3x3_wgrad(src, dzc, cout, cin, want_bias=ctx.has_bias)
            else:
                dw, db = H.conv4x4s2_wgrad(src, dzc, cout, cin, want_bias=ctx.has_bias)
        return dx, dw, (db if ctx.has_bias else None), None


class BNLReLUFn(torch.autograd.Function):
    """nn.BatchNorm2d + LeakyReLU on CB8 (sr_bn_lrelu_fwd_f32 / sr_bn_lrelu_bwd_f32)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, train, momentum, eps, slope):
        lib = _lib.load()
        n, cb, h, w, _ = x.shape
        c = gamma.numel()
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        wsb = lib.sr_reduce_workspace_bytes(c)
        ws = scratch(dev, wsb)
        ns = cb * h * w * 8
        with torch.cuda.device(dev):
            _lib.check(lib.sr_bn_lrelu_fwd_f32(x.data_ptr(), ns, y.data_ptr(), ns, n, c, h, w, gamma.data_ptr(),
                                               beta.data_ptr(), running_mean.data_ptr() if running_mean is not None else None,
                                               running_var.data_ptr() if running_var is not None else None, int(train),
                                               momentum, eps, slope, mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(),
                                               wsb, _stream(dev)), 'sr_bn_lrelu_fwd_f32')
        ctx.save_for_backward(x, y, gamma, mean, invstd)
        ctx.train, ctx.slope = bool(train), slope
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, y, gamma, mean, invstd = ctx.saved_tensors
        n, cb, h, w, _ = x.shape
        c = gamma.numel()
        dev = x.device
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        dgamma = torch.empty_like(gamma)
        dbeta = torch.empty_like(gamma)
        wsb = lib.sr_reduce_workspace_bytes(c)
        ws = scratch(dev, wsb)
        ns = cb * h * w * 8
        with torch.cuda.device(dev):
            _lib.check(lib.sr_bn_lrelu_bwd_f32(x.data_ptr(), ns, gy.data_ptr(), ns, y.data_ptr(), ns, dx.data_ptr(), ns, n, c,
                                               h, w, gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(), int(ctx.train),
                                               ctx.slope, dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), wsb,
                                               _stream(dev)), 'sr_bn_lrelu_bwd_f32')
        return dx, dgamma, dbeta, None, None, None, None, None, None


class LinearFn(torch.autograd.Function):
    """nn.Linear + LeakyReLU(act_slope) (sr_linear_fwd_f32 / sr_linear_bwd_f32)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act_slope):
        lib = _lib.load()
        x = x.contiguous()
        n, nin = x.shape
        nout = weight.size(0)
        y = torch.empty((n, nout), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_linear_fwd_f32(x.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                             y.data_ptr(), n, nin, nout, act_slope, _stream(x.device)), 'sr_linear_fwd_f32')
        ctx.save_for_backward(x, weight, y)
        ctx.act_slope, ctx.has_bias = act_slope, bias is not None
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, weight, y = ctx.saved_tensors
        n, nin = x.shape
        nout = weight.size(0)
        gy = gy.contiguous()
        dev = x.device
        dz = torch.empty_like(gy)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        db = torch.empty(nout, dtype=torch.float32, device=dev) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        if db is not None and dw is None:
            dw = torch.empty_like(weight)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_linear_bwd_f32(x.data_ptr(), weight.data_ptr(), y.data_ptr(), gy.data_ptr(), n, nin, nout,
                                             ctx.act_slope, dz.data_ptr(), dx.data_ptr() if dx is not None else None,
                                             dw.data_ptr() if dw is not None else None,
                                             db.data_ptr() if db is not None else None, _stream(dev)), 'sr_linear_bwd_f32')
        return dx, (dw if ctx.needs_input_grad[1] else None), db, None


class L1LossFn(torch.autograd.Function):
    """loss_weight * mean|pred - target| (sr_l1_loss_fwd_f32 / sr_l1_loss_bwd_f32)."""

    @staticmethod
    def forward(ctx, pred, target, weight):
        lib = _lib.load()
        pred, target = pred.contiguous(), target.contiguous()
        dev = pred.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        wsb = lib.sr_reduce_workspace_bytes(8)
        ws = scratch(dev, wsb)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_l1_loss_fwd_f32(pred.data_ptr(), target.data_ptr(), pred.numel(), weight, loss.data_ptr(),
                                              ws.data_ptr(), wsb, _stream(dev)), 'sr_l1_loss_fwd_f32')
        ctx.save_for_backward(pred, target)
        ctx.weight = weight
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        pred, target = ctx.saved_tensors
        dev = pred.device
        g = g.contiguous().float()
        dp = torch.empty_like(pred)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_l1_loss_bwd_f32(pred.data_ptr(), target.data_ptr(), pred.numel(), ctx.weight, g.data_ptr(),
                                              dp.data_ptr(), _stream(dev)), 'sr_l1_loss_bwd_f32')
        return dp, None, None


class GramFn(torch.autograd.Function):
    """[n, c, h, w] -> [n, c, c] = F F^T / (c h w), F = the feature map as [c, h w] (PerceptualLoss._gram_mat, losses.py:342-356):
    sr_gram_fwd_f32 / sr_gram_bwd_f32."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous().float()
        n, c, h, w = x.shape
        dev = x.device
        g = torch.empty((n, c, c), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_gram_fwd_f32(x.data_ptr(), n, c, h * w, 1.0 / (c * h * w), g.data_ptr(), _stream(dev)), 'sr_gram_fwd_f32')
        ctx.save_for_backward(x)
        return g

    @staticmethod
    def backward(ctx, dg):
        lib = _lib.load()
        x, = ctx.saved_tensors
        n, c, h, w = x.shape
        dev = x.device
        dx = torch.empty_like(x)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_gram_bwd_f32(x.data_ptr(), dg.contiguous().float().data_ptr(), n, c, h * w, 1.0 / (c * h * w), dx.data_ptr(),
                                           _stream(dev)), 'sr_gram_bwd_f32')
        return dx


class PixelLossFn(torch.autograd.Function):
    """weight * mean(criterion(pred - target)), criterion kind 1 = squared error, 2 = Charbonnier(eps)
    (sr_pixel_loss_fwd_f32 / sr_pixel_loss_bwd_f32)."""

    @staticmethod
    def forward(ctx, pred, target, weight, kind, eps):
        lib = _lib.load()
        pred, target = pred.contiguous(), target.contiguous()
        dev = pred.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        wsb = lib.sr_reduce_workspace_bytes(8)
        ws = scratch(dev, wsb)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_pixel_loss_fwd_f32(pred.data_ptr(), target.data_ptr(), pred.numel(), kind, eps, weight,
                                                 loss.data_ptr(), ws.data_ptr(), wsb, _stream(dev)), 'sr_pixel_loss_fwd_f32')
        ctx.save_for_backward(pred, target)
        ctx.args = (weight, kind, eps)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        pred, target = ctx.saved_tensors
        weight, kind, eps = ctx.args
        dev = pred.device
        dp = torch.empty_like(pred)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_pixel_loss_bwd_f32(pred.data_ptr(), target.data_ptr(), pred.numel(), kind, eps, weight,
                                                 g.contiguous().float().data_ptr(), dp.data_ptr(), _stream(dev)),
                       'sr_pixel_loss_bwd_f32')
        return dp, None, None, None, None


class GanPointLossFn(torch.autograd.Function):
    """weight * mean f(x) for the point-wise GAN criteria (sr_gan_point_loss_{fwd,bwd}_f32; kinds in include/sr_hip.h)."""

    @staticmethod
    def forward(ctx, x, kind, c, weight):
        lib = _lib.load()
        x = x.contiguous().float()
        dev = x.device
        loss = torch.empty((), dtype=torch.float32, device=dev)
        wsb = lib.sr_reduce_workspace_bytes(8)
        ws = scratch(dev, wsb)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_gan_point_loss_fwd_f32(x.data_ptr(), x.numel(), kind, c, weight, loss.data_ptr(), ws.data_ptr(), wsb,
                                                     _stream(dev)), 'sr_gan_point_loss_fwd_f32')
        ctx.save_for_backward(x)
        ctx.args = (kind, c, weight)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        kind, c, weight = ctx.args
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_gan_point_loss_bwd_f32(x.data_ptr(), x.numel(), kind, c, weight, g.contiguous().float().data_ptr(),
                                                     dx.data_ptr(), _stream(x.device)), 'sr_gan_point_loss_bwd_f32')
        return dx, None, None, None


class BCELogitsFn(torch.autograd.Function):
    """weight * BCEWithLogits(x - mean(other), target) with `other` optional (plain GAN loss when None).

    The relativistic-average form of esrgan_model.py:40-41,67,71; gradients reach both x and other."""

    @staticmethod
    def forward(ctx, x, other, target_is_real, weight):
        lib = _lib.load()
        x = x.contiguous()
        dev = x.device
        wsb = lib.sr_reduce_workspace_bytes(8)
        ws = scratch(dev, wsb)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        shift = dsum = None
        with torch.cuda.device(dev):
            if other is not None:
                other = other.contiguous()
                shift = torch.empty((), dtype=torch.float32, device=dev)
                dsum = torch.empty((), dtype=torch.float32, device=dev)
                _lib.check(lib.sr_mean_f32(other.data_ptr(), other.numel(), shift.data_ptr(), ws.data_ptr(), wsb, _stream(dev)),
                           'sr_mean_f32')
            _lib.check(lib.sr_bce_logits_fwd_f32(x.data_ptr(), shift.data_ptr() if shift is not None else None, x.numel(),
                                                 int(target_is_real), weight, loss.data_ptr(),
                                                 dsum.data_ptr() if dsum is not None else None, ws.data_ptr(), wsb,
                                                 _stream(dev)), 'sr_bce_logits_fwd_f32')
        ctx.save_for_backward(x, shift, dsum)
        ctx.target_is_real, ctx.weight = bool(target_is_real), weight
        ctx.other_shape = tuple(other.shape) if other is not None else None
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, shift, dsum = ctx.saved_tensors
        dev = x.device
        g = g.contiguous().float()
        dx = dother = None
        with torch.cuda.device(dev):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _lib.check(lib.sr_bce_logits_bwd_f32(x.data_ptr(), shift.data_ptr() if shift is not None else None,
                                                     x.numel(), int(ctx.target_is_real), ctx.weight, g.data_ptr(),
                                                     dx.data_ptr(), _stream(dev)), 'sr_bce_logits_bwd_f32')
            if ctx.other_shape is not None and ctx.needs_input_grad[1]:
                dother = torch.empty(ctx.other_shape, dtype=torch.float32, device=dev)
                _lib.check(lib.sr_fill_scaled_f32(g.data_ptr(), dsum.data_ptr(), -1.0 / dother.numel(), dother.data_ptr(),
                                                  dother.numel(), _stream(dev)), 'sr_fill_scaled_f32')
        return dx, dother, None, None


def mean(x):
    """torch.mean(x.detach()) as one HIP reduction (logging of out_d_real / out_d_fake, esrgan_model.py:77-78)."""
    lib = _lib.load()
    x = x.detach().contiguous()
    dev = x.device
    out = torch.empty((), dtype=torch.float32, device=dev)
    wsb = lib.sr_reduce_workspace_bytes(8)
    ws = scratch(dev, wsb)
    with torch.cuda.device(dev):
        _lib.check(lib.sr_mean_f32(x.data_ptr(), x.numel(), out.data_ptr(), ws.data_ptr(), wsb, _stream(dev)), 'sr_mean_f32')
    return out


class Bilinear2xFn(torch.autograd.Function):
    """F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) on CB8 (sr_bilinear2x_{fwd,bwd}_f32)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        n, cb, h, w, _ = x.shape
        y = torch.empty((n, cb, 2 * h, 2 * w, 8), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_bilinear2x_fwd_f32(x.data_ptr(), cb * h * w * 8, y.data_ptr(), cb * h * w * 32, n, cb, h, w,
                                                 _stream(x.device)), 'sr_bilinear2x_fwd_f32')
        ctx.shape = (n, cb, h, w)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        n, cb, h, w = ctx.shape
        g = g.contiguous()
        gx = torch.empty((n, cb, h, w, 8), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(lib.sr_bilinear2x_bwd_f32(g.data_ptr(), cb * h * w * 32, gx.data_ptr(), cb * h * w * 8, n, cb, h, w,
                                                 _stream(g.device)), 'sr_bilinear2x_bwd_f32')
        return gx


class AddFn(torch.autograd.Function):
    """a + b of two CB8 activations (sr_add_f32); both inputs receive the incoming gradient."""

    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        with torch.cuda.device(a.device):
            _lib.check(lib.sr_add_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream(a.device)), 'sr_add_f32')
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class SpectralNormFn(torch.autograd.Function):
    """weight = weight_orig / sigma with one power iteration on (u, v) in train mode
    (sr_spectral_norm_{fwd,bwd}_f32); u and v are buffers updated in place, as torch.nn.utils.spectral_norm does."""

    @staticmethod
    def forward(ctx, weight_orig, u, v, update, eps):
        lib = _lib.load()
        w = weight_orig.contiguous()
        rows = w.size(0)
        cols = w.numel() // rows
        dev = w.device
        w_sn = torch.empty_like(w)
        sigma = torch.empty((), dtype=torch.float32, device=dev)
        wsb = max((rows + 16 * cols) * 4, lib.sr_reduce_workspace_bytes(8) + 64)
        ws = scratch(dev, wsb)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_spectral_norm_fwd_f32(w.data_ptr(), u.data_ptr(), v.data_ptr(), rows, cols, int(update), eps,
                                                    w_sn.data_ptr(), sigma.data_ptr(), ws.data_ptr(), wsb, _stream(dev)),
                       'sr_spectral_norm_fwd_f32')
        ctx.save_for_backward(w_sn, u.clone(), v.clone(), sigma)
        ctx.dims = (rows, cols)
        return w_sn

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        w_sn, u, v, sigma = ctx.saved_tensors
        rows, cols = ctx.dims
        dev = g.device
        g = g.contiguous()
        gw = torch.empty_like(g)
        wsb = max((rows + cols) * 4, lib.sr_reduce_workspace_bytes(8) + 64)
        ws = scratch(dev, wsb)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_spectral_norm_bwd_f32(g.data_ptr(), w_sn.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(),
                                                    rows, cols, gw.data_ptr(), ws.data_ptr(), wsb, _stream(dev)),
                       'sr_spectral_norm_bwd_f32')
        return gw, None, None, None, None


class SpectralNormBatchFn(torch.autograd.Function):
    """SpectralNormFn for all spectral-norm layers of a network in one call (sr_spectral_norm_fwd_batch_f32: one launch per stage
    of the power iteration for all layers): forward(update, eps, w0, u0, v0, w1, u1, v1, ...) -> (w_sn0, w_sn1, ...); per layer the
    same values as SpectralNormFn.  The backward stays per layer (one launch each)."""

    @staticmethod
    def forward(ctx, update, eps, *wuv):
        lib = _lib.load()
        nl = len(wuv) // 3
        ws_, us, vs = [w.contiguous() for w in wuv[0::3]], wuv[1::3], wuv[2::3]
        dev = ws_[0].device
        outs = [torch.empty_like(w) for w in ws_]
        sigmas = torch.empty(nl, dtype=torch.float32, device=dev)
        table = (_lib.SnLayer * nl)()
        need = 0
        dims = []
        for i, (w, u, v) in enumerate(zip(ws_, us, vs)):
            rows = w.size(0)
            cols = w.numel() // rows
            dims.append((rows, cols))
            table[i].w_orig, table[i].u, table[i].v = w.data_ptr(), u.data_ptr(), v.data_ptr()
            table[i].rows, table[i].cols = rows, cols
            table[i].w_sn, table[i].sigma = outs[i].data_ptr(), sigmas.data_ptr() + 4 * i
            need += ((rows + 16 * cols) * 4 + 255) // 256 * 256
        ws = scratch(dev, need + 256)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_spectral_norm_fwd_batch_f32(table, nl, int(update), eps, ws.data_ptr(), need + 256, _stream(dev)),
                       'sr_spectral_norm_fwd_batch_f32')
        if any(ctx.needs_input_grad[2::3]):   # u, v are updated in place by the next forward: the backward needs this forward's
            ctx.save_for_backward(sigmas, *outs, *[u.clone() for u in us], *[v.clone() for v in vs])
        ctx.dims = dims
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gs):
        lib = _lib.load()
        saved = ctx.saved_tensors
        nl = len(ctx.dims)
        sigmas, outs, us, vs = saved[0], saved[1:1 + nl], saved[1 + nl:1 + 2 * nl], saved[1 + 2 * nl:1 + 3 * nl]
        grads = [None, None]
        for i, g in enumerate(gs):
            if g is None:
                grads += [None, None, None]
                continue
            rows, cols = ctx.dims[i]
            dev = g.device
            g = g.contiguous()
            gw = torch.empty_like(g)
            wsb = max((rows + cols) * 4, lib.sr_reduce_workspace_bytes(8) + 64)
            ws = scratch(dev, wsb)
            with torch.cuda.device(dev):
                _lib.check(lib.sr_spectral_norm_bwd_f32(g.data_ptr(), outs[i].data_ptr(), us[i].data_ptr(), vs[i].data_ptr(),
                                                        sigmas.data_ptr() + 4 * i, rows, cols, gw.data_ptr(), ws.data_ptr(), wsb,
                                                        _stream(dev)), 'sr_spectral_norm_bwd_f32')
            grads += [gw, None, None]
        return tuple(grads)


class MaxPool2x2Fn(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=2, stride=2) on CB8 (sr_maxpool2x2_fwd_f32 / sr_maxpool2x2_bwd_f32)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous()
        n, cb, h, w, _ = x.shape
        y = torch.empty((n, cb, h // 2, w // 2, 8), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_maxpool2x2_fwd_f32(x.data_ptr(), y.data_ptr(), n, cb, h, w, _stream(x.device)), 'sr_maxpool2x2_fwd_f32')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, = ctx.saved_tensors
        n, cb, h, w, _ = x.shape
        dx = torch.empty_like(x)
        gy = gy.contiguous()
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_maxpool2x2_bwd_f32(x.data_ptr(), gy.data_ptr(), dx.data_ptr(), n, cb, h, w, _stream(x.device)),
                       'sr_maxpool2x2_bwd_f32')
        return dx


class ChannelAffineFn(torch.autograd.Function):
    """y[n][c] = x[n][c] * a[c] + b[c] on NCHW fp32 (constants a, b: the VGG input normalisation) — sr_channel_affine_f32."""

    @staticmethod
    def forward(ctx, x, a, b):
        lib = _lib.load()
        x = x.contiguous().float()
        n, c, h, w = x.shape
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_channel_affine_f32(x.data_ptr(), y.data_ptr(), a.data_ptr(), b.data_ptr(), n, c, h * w, _stream(x.device)),
                       'sr_channel_affine_f32')
        ctx.save_for_backward(a)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        a, = ctx.saved_tensors
        gy = gy.contiguous()
        n, c, h, w = gy.shape
        dx = torch.empty_like(gy)
        with torch.cuda.device(gy.device):
            _lib.check(lib.sr_channel_affine_f32(gy.data_ptr(), dx.data_ptr(), a.data_ptr(), None, n, c, h * w, _stream(gy.device)),
                       'sr_channel_affine_f32')
        return dx, None, None


class LReLUFn(torch.autograd.Function):
    """Stand-alone LeakyReLU(slope) / ReLU (slope 0) on any contiguous fp32 tensor (sr_lrelu_fwd_f32 / sr_lrelu_bwd_f32)."""

    @staticmethod
    def forward(ctx, x, slope):
        lib = _lib.load()
        x = x.contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_lrelu_fwd_f32(x.data_ptr(), y.data_ptr(), slope, x.numel(), _stream(x.device)), 'sr_lrelu_fwd_f32')
        ctx.save_for_backward(y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        y, = ctx.saved_tensors
        gy = gy.contiguous()
        dx = torch.empty_like(gy)
        with torch.cuda.device(gy.device):
            _lib.check(lib.sr_lrelu_bwd_f32(gy.data_ptr(), y.data_ptr(), dx.data_ptr(), ctx.slope, gy.numel(), _stream(gy.device)),
                       'sr_lrelu_bwd_f32')
        return dx, None

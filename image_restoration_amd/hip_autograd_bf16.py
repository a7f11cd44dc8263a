"""torch.autograd.Function wrappers of the bf16 single-op C ABI (discriminator with compute_dtype='bf16').

Activations travel between these Functions as torch.bfloat16 tensors shaped [N, C/16, H, W, 16] (the CB16 layout of
include/sr_hip.h); weights, biases and their gradients stay fp32 (the master copies the optimiser updates); every
forward / backward below is libsr_hip.so launches only.  Twin of hip_autograd.py (fp32, CB8).
"""
import ctypes as C

import torch

from . import _lib
from . import hip_ops as H


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _cb16(t):
    return H.CB16(t)


class ToCB16(torch.autograd.Function):
    """NCHW fp32 -> CB16 bf16 (sr_nchw_to_cb16_bf16); backward CB16 -> NCHW fp32."""

    @staticmethod
    def forward(ctx, x):
        ctx.c = x.size(1)
        return H.nchw_to_cb16(x).buf

    @staticmethod
    def backward(ctx, g):
        return H.cb16_to_nchw(_cb16(g.contiguous()), ctx.c)


def _unshuffle2(t, inverse=False):
    """CB16 [N, C/16, 2h, 2w, 16] <-> [N, 4C/16, h, w, 16] (parity-major channels) — sr_cb16_unshuffle2_bf16."""
    lib = _lib.load()
    n, cb, hh, ww, _ = t.shape
    if inverse:
        cblocks, h, w = cb // 4, hh, ww
        out = torch.empty((n, cblocks, 2 * h, 2 * w, 16), dtype=torch.bfloat16, device=t.device)
    else:
        cblocks, h, w = cb, hh // 2, ww // 2
        out = torch.empty((n, 4 * cblocks, h, w, 16), dtype=torch.bfloat16, device=t.device)
    with torch.cuda.device(t.device):
        _lib.check(lib.sr_cb16_unshuffle2_bf16(t.data_ptr(), t[0].numel(), out.data_ptr(), out[0].numel(), n, cblocks, h, w,
                                               int(inverse), _stream(t.device)), 'sr_cb16_unshuffle2_bf16')
    return out


def _w4_as_w3(w4):
    lib = _lib.load()
    cout, cin = w4.shape[:2]
    w3 = torch.empty((cout, 4 * cin, 3, 3), dtype=torch.float32, device=w4.device)
    with torch.cuda.device(w4.device):
        _lib.check(lib.sr_conv4x4s2_weight_as_3x3_f32(w4.data_ptr(), w3.data_ptr(), cout, cin, 0, _stream(w4.device)),
                   'sr_conv4x4s2_weight_as_3x3_f32')
    return w3


def _dw3_to_dw4(dw3, cout, cin):
    lib = _lib.load()
    dw4 = torch.empty((cout, cin, 4, 4), dtype=torch.float32, device=dw3.device)
    with torch.cuda.device(dw3.device):
        _lib.check(lib.sr_conv4x4s2_weight_as_3x3_f32(dw4.data_ptr(), dw3.data_ptr(), cout, cin, 1, _stream(dw3.device)),
                   'sr_conv4x4s2_weight_as_3x3_f32')
    return dw4


class ConvFn16(torch.autograd.Function):
    """3x3/s1/p1 or 4x4/s2/p1 convolution (+bias, +LeakyReLU(act_slope)) on CB16 with fp32 master weights.

    forward  : sr_conv3x3_bf16; a 4x4/s2 conv = the same kernel on the pixel-unshuffled input with the 3x3-embedded
               weights (csrc/disc_bf16.hip).  out_nchw=True returns fp32 NCHW (the logit map of the last layer).
    backward : sr_lrelu_bwd_bf16, data gradient = sr_conv3x3_bf16 with the transposed image (+ pixel shuffle back for 4x4),
               weight gradient = sr_conv3x3_wgrad_bf16 in fp32 (folded back to 4x4).

    Three options let neighbouring layers share passes (the caller guarantees the stated conditions):
      pre_unshuffled   the input of a 4x4 conv already is the pixel-unshuffled tensor (SkipForkFn16 made it); the returned
                       gradient stays in that layout;
      input_slope      != 1: the input is the LeakyReLU(input_slope) output of a layer with no other consumer, so the data
                       gradient is multiplied by that LeakyReLU's derivative in the conv's epilogue (mask = the input) and
                       the producer must be built with grad_premasked=True;
      grad_premasked   the gradient arriving for this conv's output already carries its LeakyReLU derivative;
      out_unshuffled   the output is stored pixel-unshuffled ONLY ([N, 4C/16, H/2, W/2, 16]: what the next 4x4/s2 conv reads;
                       sr_conv3x3_desc.out_unshuffle2) and handed to ForkU2Fn16.  Convention for such tensors: their GRADIENT
                       travels in the plain layout ([N, C/16, H, W, 16] bytes) under the unshuffled shape — same element count, and
                       every producer / consumer of it is one of this module's functions;
      skip_u2          a skip connection that only exists pixel-unshuffled (ForkU2Fn16's handle) is added to the activation in the
                       conv's epilogue: out = LeakyReLU(conv) + skip, stored with the sign-keeping rounding of
                       sr_conv3x3_desc.res1_keep_sign, so neither the activation nor a separate sum pass exists; the backward
                       recovers the LeakyReLU mask from (out, skip) exactly (sr_lrelu_bwd_diff_u2_bf16) and hands the skip the
                       plain gradient (under the unshuffled shape, as above).
    """

    @staticmethod
    def forward(ctx, x, weight, bias, act_slope, out_nchw, pre_unshuffled=False, input_slope=1.0, grad_premasked=False,
                out_unshuffled=False, skip_u2=None):
        k = weight.size(2)
        param = weight   # (the cache below keys on the parameter object)
        weight = weight.detach().contiguous().float()
        cout, cin = weight.shape[:2]
        if k == 4:
            src = _cb16(x.contiguous() if pre_unshuffled else _unshuffle2(x.contiguous()))
        elif k == 3:
            src = _cb16(x.contiguous())
        else:
            raise NotImplementedError(f'kernel size {k}')

        def build():
            w3_ = _w4_as_w3(weight) if k == 4 else weight
            return w3_, H.PackedConvBF16(w3_, bias)
        w3, pc = H.cached_pack('bf16 fwd', param, bias, build)
        s2 = cin if (k == 4 and cin % 64 == 0) else 0  # lets the kernel skip the zero taps of the 3x3 embedding
        if out_nchw:
            assert act_slope == 1.0
            y = torch.empty((src.n, cout, src.h, src.w), dtype=torch.float32, device=x.device)
            H.conv3x3_bf16(src, pc, out_nchw=y)
            saved_y = None
            ret = y
        elif skip_u2 is not None:
            assert k == 3 and act_slope != 1.0 and not grad_premasked and not out_unshuffled
            skip_u2 = skip_u2.contiguous()
            assert skip_u2.shape == (src.n, 4 * cout // 16, src.h // 2, src.w // 2, 16), (tuple(skip_u2.shape), src.n, cout, src.h, src.w)
            out = H.conv3x3_bf16(src, pc, act_slope=act_slope, res1=_cb16(skip_u2), beta1=1.0, res1_u2=True, res1_keep_sign=True)
            saved_y = out.buf     # = activation + skip: the mask is sign(saved_y - skip)
            ret = out.buf
        else:
            out = H.conv3x3_bf16(src, pc, act_slope=act_slope, s2_channels=s2, out_unshuffle2=out_unshuffled)
            assert not out_unshuffled or grad_premasked, 'the fork applies the LeakyReLU derivative of an unshuffled output'
            saved_y = out.buf if (act_slope != 1.0 and not grad_premasked) else None
            ret = out.buf
        ctx.save_for_backward(src.buf, w3, saved_y, skip_u2 if (skip_u2 is not None and not out_nchw) else None)
        ctx.act_slope, ctx.has_bias, ctx.k, ctx.out_nchw = act_slope, bias is not None, k, out_nchw
        ctx.cout, ctx.cin, ctx.x_cb = cout, cin, x.size(1)
        ctx.pre_unshuffled, ctx.input_slope, ctx.grad_premasked, ctx.s2 = pre_unshuffled, input_slope, grad_premasked, s2
        ctx.out_unshuffled = out_unshuffled
        ctx.param = param if isinstance(param, torch.nn.Parameter) else None   # key of the transposed image's cache entry
        assert input_slope == 1.0 or k == 3, 'the input mask applies to 3x3 convs'
        return ret

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        xs, w3, y, skip_u2 = ctx.saved_tensors
        src = _cb16(xs)
        dev = gy.device
        cout, cin3 = w3.shape[:2]
        if ctx.out_nchw:
            dzc = H.nchw_to_cb16(gy.contiguous().float())
        else:
            gy = gy.contiguous()
            if ctx.out_unshuffled:  # plain-layout gradient under the unshuffled shape (see the class docstring)
                n_, cb4, hh, ww, _ = gy.shape
                gy = gy.view(n_, cb4 // 4, 2 * hh, 2 * ww, 16)
            if skip_u2 is not None:     # y = activation + skip (sign-keeping rounding): mask = sign(y - skip)
                dz = torch.empty_like(gy)
                n_, cb_, h2_, w2_, _ = gy.shape
                with torch.cuda.device(dev):
                    _lib.check(lib.sr_lrelu_bwd_diff_u2_bf16(gy.data_ptr(), y.data_ptr(), skip_u2.data_ptr(), dz.data_ptr(), ctx.act_slope,
                                                             n_, cb_, h2_ // 2, w2_ // 2, _stream(dev)), 'sr_lrelu_bwd_diff_u2_bf16')
            elif y is not None and not ctx.grad_premasked:
                dz = torch.empty_like(gy)
                with torch.cuda.device(dev):
                    _lib.check(lib.sr_lrelu_bwd_bf16(gy.data_ptr(), y.data_ptr(), dz.data_ptr(), ctx.act_slope, gy.numel(),
                                                     _stream(dev)), 'sr_lrelu_bwd_bf16')
            else:
                dz = gy
            dzc = _cb16(dz)
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dx = dw = db = None
        if need_x:
            pct = H.cached_pack('bf16 dgrad', ctx.param, None, lambda: H.PackedConvBF16(w3, None, mode=1))
            if ctx.input_slope != 1.0:  # dL/d(pre-activation of the producer): mask = this conv's own input
                d = H.conv3x3_bf16(dzc, pct, mask=src, mask_slope=ctx.input_slope).buf
            else:
                d = H.conv3x3_bf16(dzc, pct, s2_channels=ctx.s2, s2_side=1).buf  # cin3 channels
            dx = _unshuffle2(d, inverse=True) if ctx.k == 4 and not ctx.pre_unshuffled else d
            if dx.size(1) != ctx.x_cb:
                dx = dx[:, :ctx.x_cb].contiguous()
        if need_w or (need_b and ctx.has_bias):
            dw, db = H.conv3x3_wgrad_bf16(src, dzc, cout, cin3, want_bias=ctx.has_bias)
            if ctx.k == 4:
                dw = _dw3_to_dw4(dw, ctx.cout, ctx.cin)
        g_skip = gy.view(skip_u2.shape) if (skip_u2 is not None and ctx.needs_input_grad[9]) else None  # plain layout, unshuffled shape
        return dx, dw, (db if ctx.has_bias else None), None, None, None, None, None, None, g_skip


class SkipForkFn16(torch.autograd.Function):
    """An encoder activation x = LeakyReLU(conv(..)) that feeds a skip connection and the next 4x4/s2 conv:
    forward returns (x, pixel_unshuffle(x)) (sr_cb16_unshuffle2_bf16); backward folds the gradient sum, the way back
    through the unshuffle and x's LeakyReLU derivative into one pass (sr_cb16_fork_bwd_bf16), so the conv that produced
    x must be built with grad_premasked=True."""

    @staticmethod
    def forward(ctx, x, slope):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.slope = slope
        ctx.set_materialize_grads(False)
        return x.view_as(x), _unshuffle2(x)

    @staticmethod
    def backward(ctx, g_skip, g_u):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        if g_u is None:  # the strided conv took no gradient (frozen network input): plain LeakyReLU backward of the skip
            g_u = torch.zeros((x.size(0), 4 * x.size(1), x.size(2) // 2, x.size(3) // 2, 16), dtype=x.dtype, device=x.device)
        n, cb, hh, ww, _ = x.shape
        dz = torch.empty_like(x)
        g_skip = g_skip.contiguous() if g_skip is not None else None
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_cb16_fork_bwd_bf16(g_skip.data_ptr() if g_skip is not None else None, g_u.contiguous().data_ptr(),
                                                 x.data_ptr(), dz.data_ptr(), ctx.slope, n, cb, hh // 2, ww // 2, _stream(x.device)),
                       'sr_cb16_fork_bwd_bf16')
        return dz, None


class ForkU2Fn16(torch.autograd.Function):
    """An encoder activation that only exists pixel-unshuffled (ConvFn16(out_unshuffled=True)) and feeds a skip connection and
    the next 4x4/s2 conv: forward hands the same tensor to both (no kernel, no copy); backward is SkipForkFn16's one pass with the
    LeakyReLU mask read from the unshuffled tensor (sr_cb16_fork_bwd_u2_bf16).  Gradients of such tensors travel in the PLAIN
    layout under the unshuffled shape (ConvFn16 docstring): g_skip arrives that way from Bilinear2xFn16 / AddFn16, the result
    leaves that way for the producing conv; g_u is the strided conv's genuine unshuffled gradient."""

    @staticmethod
    def forward(ctx, u, slope):
        u = u.contiguous()
        ctx.save_for_backward(u)
        ctx.slope = slope
        ctx.set_materialize_grads(False)
        return u.view_as(u), u.view_as(u)

    @staticmethod
    def backward(ctx, g_skip, g_u):
        lib = _lib.load()
        (u,) = ctx.saved_tensors
        n, cb4, h, w, _ = u.shape
        if g_u is None:
            g_u = torch.zeros_like(u)
        dz = torch.empty_like(u)   # plain layout [n, cb4 / 4, 2h, 2w, 16] under u's shape
        g_skip = g_skip.contiguous() if g_skip is not None else None
        with torch.cuda.device(u.device):
            _lib.check(lib.sr_cb16_fork_bwd_u2_bf16(g_skip.data_ptr() if g_skip is not None else None, g_u.contiguous().data_ptr(),
                                                    u.data_ptr(), dz.data_ptr(), ctx.slope, n, cb4 // 4, h, w, _stream(u.device)),
                       'sr_cb16_fork_bwd_u2_bf16')
        return dz, None


class Bilinear2xFn16(torch.autograd.Function):
    """F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) on CB16 (sr_bilinear2x_{fwd,bwd}_bf16); with a
    second input the resampled tensor is x + skip (the skip connection folded into the same pass).

    input_slope != 1: x is the LeakyReLU(input_slope) output of a conv whose only consumer this is; the backward then returns
    dL/d(that conv's pre-activation) for x (sr_bilinear2x_bwd_lrelu_bf16: the derivative rides on the resampling gradient's
    store) and the conv must be built with grad_premasked=True.  The skip input always receives the plain gradient.
    skip_u2: the skip input only exists pixel-unshuffled (ForkU2Fn16); it is read where it is (sr_bilinear2x_fwd_u2_bf16) and its
    gradient goes back in the plain layout under the unshuffled shape."""

    @staticmethod
    def forward(ctx, x, skip=None, input_slope=1.0, skip_u2=False):
        lib = _lib.load()
        x = x.contiguous()
        n, cb, h, w, _ = x.shape
        y = torch.empty((n, cb, 2 * h, 2 * w, 16), dtype=torch.bfloat16, device=x.device)
        if skip is not None:
            skip = skip.contiguous()
            assert skip.shape == ((n, 4 * cb, h // 2, w // 2, 16) if skip_u2 else x.shape)
        with torch.cuda.device(x.device):
            if skip is not None and skip_u2:
                _lib.check(lib.sr_bilinear2x_fwd_u2_bf16(x.data_ptr(), x[0].numel(), skip.data_ptr(), skip[0].numel(), y.data_ptr(),
                                                         y[0].numel(), n, cb, h, w, _stream(x.device)), 'sr_bilinear2x_fwd_u2_bf16')
            else:
                _lib.check(lib.sr_bilinear2x_fwd_bf16(x.data_ptr(), x[0].numel(), skip.data_ptr() if skip is not None else None,
                                                      skip[0].numel() if skip is not None else 0, y.data_ptr(), y[0].numel(), n, cb, h,
                                                      w, _stream(x.device)), 'sr_bilinear2x_fwd_bf16')
        ctx.has_skip = skip is not None
        ctx.skip_shape = tuple(skip.shape) if skip is not None else None
        ctx.input_slope = input_slope
        if input_slope != 1.0:
            ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        g = g.contiguous()
        n, cb, h2, w2, _ = g.shape
        gx = torch.empty((n, cb, h2 // 2, w2 // 2, 16), dtype=torch.bfloat16, device=g.device)
        if ctx.input_slope != 1.0:
            (x,) = ctx.saved_tensors
            want_plain = ctx.has_skip and ctx.needs_input_grad[1]
            gplain = torch.empty_like(gx) if want_plain else None
            with torch.cuda.device(g.device):
                _lib.check(lib.sr_bilinear2x_bwd_lrelu_bf16(g.data_ptr(), g[0].numel(), gx.data_ptr(), gx[0].numel(), x.data_ptr(),
                                                            x[0].numel(), ctx.input_slope, gplain.data_ptr() if want_plain else None,
                                                            gplain[0].numel() if want_plain else 0, n, cb, h2 // 2, w2 // 2,
                                                            _stream(g.device)), 'sr_bilinear2x_bwd_lrelu_bf16')
            return gx, (gplain.view(ctx.skip_shape) if want_plain else None), None, None
        with torch.cuda.device(g.device):
            _lib.check(lib.sr_bilinear2x_bwd_bf16(g.data_ptr(), g[0].numel(), gx.data_ptr(), gx[0].numel(), n, cb, h2 // 2, w2 // 2,
                                                  _stream(g.device)), 'sr_bilinear2x_bwd_bf16')
        return gx, (gx.view(ctx.skip_shape) if ctx.has_skip else None), None, None


class AddFn16(torch.autograd.Function):
    """a + b on CB16 (skip connections) in one pass: sr_cb16_add_bf16; b_u2: b only exists pixel-unshuffled (ForkU2Fn16) and is read
    where it is (sr_cb16_add_u2_bf16), its gradient goes back in the plain layout under the unshuffled shape."""

    @staticmethod
    def forward(ctx, a, b, b_u2=False):
        lib = _lib.load()
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        ctx.b_shape = tuple(b.shape)
        with torch.cuda.device(a.device):
            if b_u2:
                n, cb, h2, w2, _ = a.shape
                assert b.shape == (n, 4 * cb, h2 // 2, w2 // 2, 16)
                _lib.check(lib.sr_cb16_add_u2_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), n, cb, h2 // 2, w2 // 2, _stream(a.device)),
                           'sr_cb16_add_u2_bf16')
            else:
                assert a.shape == b.shape
                _lib.check(lib.sr_cb16_add_bf16(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream(a.device)),
                           'sr_cb16_add_bf16')
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g.view(ctx.b_shape), None


class MaxPool2x2Fn16(torch.autograd.Function):
    """nn.MaxPool2d(kernel_size=2, stride=2) on CB16 (sr_maxpool2x2_{fwd,bwd}_bf16)."""

    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x = x.contiguous()
        n, cb, h, w, _ = x.shape
        y = torch.empty((n, cb, h // 2, w // 2, 16), dtype=torch.bfloat16, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_maxpool2x2_fwd_bf16(x.data_ptr(), y.data_ptr(), n, cb, h, w, _stream(x.device)), 'sr_maxpool2x2_fwd_bf16')
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        n, cb, h, w, _ = x.shape
        dx = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_maxpool2x2_bwd_bf16(x.data_ptr(), g.contiguous().data_ptr(), dx.data_ptr(), n, cb, h, w,
                                                  _stream(x.device)), 'sr_maxpool2x2_bwd_bf16')
        return dx


class LReLUFn16(torch.autograd.Function):
    """Stand-alone LeakyReLU / ReLU on CB16 (sr_lrelu_fwd_bf16 / sr_lrelu_bwd_bf16)."""

    @staticmethod
    def forward(ctx, x, slope):
        lib = _lib.load()
        x = x.contiguous()
        y = torch.empty_like(x)
        with torch.cuda.device(x.device):
            _lib.check(lib.sr_lrelu_fwd_bf16(x.data_ptr(), y.data_ptr(), slope, x.numel(), _stream(x.device)), 'sr_lrelu_fwd_bf16')
        ctx.save_for_backward(y)
        ctx.slope = slope
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        dz = torch.empty_like(g)
        with torch.cuda.device(g.device):
            _lib.check(lib.sr_lrelu_bwd_bf16(g.data_ptr(), y.data_ptr(), dz.data_ptr(), ctx.slope, g.numel(), _stream(g.device)),
                       'sr_lrelu_bwd_bf16')
        return dz, None


class FromCB16(torch.autograd.Function):
    """CB16 bf16 -> NCHW fp32 with `channels` real channels; backward NCHW fp32 -> CB16 (pad channels zero)."""

    @staticmethod
    def forward(ctx, t, channels):
        ctx.cb = t.size(1)
        return H.cb16_to_nchw(_cb16(t.contiguous()), channels)

    @staticmethod
    def backward(ctx, g):
        out = H.nchw_to_cb16(g.contiguous().float())
        assert out.buf.size(1) == ctx.cb
        return out.buf, None


class BNLReLUFn16(torch.autograd.Function):
    """nn.BatchNorm2d + LeakyReLU on CB16 (sr_bn_lrelu_fwd_bf16 / sr_bn_lrelu_bwd_bf16); parameters, statistics and running
    buffers fp32."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, train, momentum, eps, slope):
        lib = _lib.load()
        x = x.contiguous()
        n, cb, h, w, _ = x.shape
        c = gamma.numel()
        dev = x.device
        y = torch.empty_like(x)
        mean = torch.empty(c, dtype=torch.float32, device=dev)
        invstd = torch.empty(c, dtype=torch.float32, device=dev)
        wsb = lib.sr_reduce_workspace_bytes(c)
        ws = H.scratch(dev, wsb)
        ns = cb * h * w * 16
        with torch.cuda.device(dev):
            _lib.check(lib.sr_bn_lrelu_fwd_bf16(x.data_ptr(), ns, y.data_ptr(), ns, n, c, h, w, gamma.data_ptr(), beta.data_ptr(),
                                                running_mean.data_ptr() if running_mean is not None else None,
                                                running_var.data_ptr() if running_var is not None else None, int(train), momentum,
                                                eps, slope, mean.data_ptr(), invstd.data_ptr(), ws.data_ptr(), wsb, _stream(dev)),
                       'sr_bn_lrelu_fwd_bf16')
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
        ws = H.scratch(dev, wsb)
        ns = cb * h * w * 16
        with torch.cuda.device(dev):
            _lib.check(lib.sr_bn_lrelu_bwd_bf16(x.data_ptr(), ns, gy.data_ptr(), ns, y.data_ptr(), ns, dx.data_ptr(), ns, n, c, h, w,
                                                gamma.data_ptr(), mean.data_ptr(), invstd.data_ptr(), int(ctx.train), ctx.slope,
                                                dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), wsb, _stream(dev)),
                       'sr_bn_lrelu_bwd_bf16')
        return dx, dgamma, dbeta, None, None, None, None, None, None

"""Thin torch-tensor wrappers over the C ABI (include/sr_hip.h).

Tensors only provide device memory and the current HIP stream; every op below is one
libsr_hip.so call.  Activations between ops are in the CB8 layout
``[N][C/8][H][W][8]`` (class ``CB8``); channel slices are views (pointer + parent stride).
"""
import ctypes as C
import weakref

import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


_scratch = {}

# ---- weight images of long-lived parameters are packed once per change, not once per call ----
# The discriminators and the perceptual extractor run their convs one autograd function at a time; every call packed its weight
# (and, in the backward, the transposed image) again: the ESRGAN step runs the discriminator five times on the same weights.
# A cached image is valid while the parameter object, its storage address, its torch version counter and its write epoch are the
# same.  The epoch covers writers torch cannot see: the fused Adam / EMA kernels write through raw pointers (optim.FlatAdam tags
# its parameters with its epoch cell and bumps it per step); invalidate_packs() is the global form.  Only nn.Parameter weights
# are cached: temporaries (a spectrally normalised weight is a new tensor every forward) would pin memory and can alias addresses.
# (An in-place write through ``param.data`` is invisible to all of these: call invalidate_packs() after one.)
_pack_epoch = [0]
_pack_cache = {}


def invalidate_packs():
    _pack_epoch[0] += 1


def cached_pack(kind, weight, bias, build):
    if not isinstance(weight, torch.nn.Parameter):
        return build()
    sig = (weight.data_ptr(), weight._version, getattr(weight, '_sr_epoch', _pack_epoch)[0], _pack_epoch[0],
           None if bias is None else (bias.data_ptr(), bias._version))
    wid = id(weight)
    slot = _pack_cache.get(wid)
    if slot is None or slot[0]() is not weight:
        # the entry (and the device memory of its images) goes when the parameter does
        slot = (weakref.ref(weight, lambda _, wid=wid: _pack_cache.pop(wid, None)), {})
        _pack_cache[wid] = slot
    hit = slot[1].get(kind)
    if hit is not None and hit[0] == sig:
        return hit[1]
    val = build()
    slot[1][kind] = (sig, val)
    return val


def scratch(dev, nbytes, tag='ws'):
    """Grow-only per-device scratch buffer (reduction workspaces, wgrad slabs).  Launches that share it are
    ordered on the current stream."""
    key = (str(dev), tag)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
        _scratch[key] = buf
    return buf


def _need_cuda(t, what):
    if not t.is_cuda:
        raise _lib.SrHipError(f'{what}: tensor is on {t.device}; the HIP path has no CPU fallback')


class CB8:
    """A channel-blocked fp32 activation: storage ``buf`` [N, CB, H, W, 8] plus a channel-block window."""

    def __init__(self, buf, cb0=0, cbn=None):
        assert buf.dim() == 5 and buf.size(4) == 8 and buf.dtype == torch.float32 and buf.is_contiguous()
        self.buf, self.cb0 = buf, cb0
        self.cbn = buf.size(1) - cb0 if cbn is None else cbn
        assert 0 <= cb0 and cb0 + self.cbn <= buf.size(1)

    @staticmethod
    def empty(n, channels, h, w, device):
        return CB8(torch.empty((n, (channels + 7) // 8, h, w, 8), dtype=torch.float32, device=device))

    @staticmethod
    def zeros(n, channels, h, w, device):
        return CB8(torch.zeros((n, (channels + 7) // 8, h, w, 8), dtype=torch.float32, device=device))

    n = property(lambda s: s.buf.size(0))
    h = property(lambda s: s.buf.size(2))
    w = property(lambda s: s.buf.size(3))
    channels = property(lambda s: s.cbn * 8)
    img_stride = property(lambda s: s.buf.size(1) * s.buf.size(2) * s.buf.size(3) * 8)
    device = property(lambda s: s.buf.device)

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.cb0 * self.h * self.w * 8 * 4

    def slice(self, c0, c):
        assert c0 % 8 == 0 and c % 8 == 0
        return CB8(self.buf, self.cb0 + c0 // 8, c // 8)


def nchw_to_cb8(x, unshuffle=1, out=None):
    _need_cuda(x, 'nchw_to_cb8')
    lib = _lib.load()
    x = x.contiguous().float()
    n, c, hh, ww = x.shape
    h, w = hh // unshuffle, ww // unshuffle
    cu = c * unshuffle * unshuffle
    if out is None:
        out = CB8.empty(n, cu, h, w, x.device)
    with torch.cuda.device(x.device):
        _lib.check(lib.sr_nchw_to_cb8_f32(x.data_ptr(), out.ptr, n, c, h, w, unshuffle, out.cbn, out.img_stride,
                                          _stream(x.device)), 'sr_nchw_to_cb8_f32')
    return out


def cb8_to_nchw(t, channels):
    lib = _lib.load()
    y = torch.empty((t.n, channels, t.h, t.w), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        _lib.check(lib.sr_cb8_to_nchw_f32(t.ptr, t.img_stride, y.data_ptr(), t.n, channels, t.h, t.w, 1,
                                          _stream(t.device)), 'sr_cb8_to_nchw_f32')
    return y


class PackedConv:
    """MFMA operand image of one 3x3 conv (sr_conv3x3_pack_f32)."""

    def __init__(self, weight, bias=None, first_seg=None, seg=0, mode=0):
        _need_cuda(weight, 'PackedConv')
        lib = _lib.load()
        weight = weight.detach().contiguous().float()
        cout, cin = weight.shape[:2]
        assert weight.shape[2:] == (3, 3)
        first_seg = cin if first_seg is None else first_seg
        self.cin_pad = lib.sr_conv3x3_cin_pad(cin, first_seg, seg)
        if self.cin_pad <= 0:
            raise ValueError(f'cin={cin} is not first_seg={first_seg} + k*seg={seg}')
        self.mode = mode
        if mode == 0:
            self.cout, self.src_channels = cout, self.cin_pad
            nw = lib.sr_conv3x3_packed_weight_floats(cout, self.cin_pad)
        else:
            self.cout, self.src_channels = self.cin_pad, (cout + 7) // 8 * 8
            nw = lib.sr_conv3x3_packed_weight_floats(self.cin_pad, self.src_channels)
        dev = weight.device
        self.w = torch.empty(nw, dtype=torch.float32, device=dev)
        self.b = None
        if mode == 0 and bias is not None:
            bias = bias.detach().contiguous().float()
            self.b = torch.empty(lib.sr_conv3x3_packed_bias_floats(cout), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_conv3x3_pack_f32(weight.data_ptr(), bias.data_ptr() if self.b is not None else None,
                                               cout, cin, first_seg, seg, mode, self.w.data_ptr(),
                                               self.b.data_ptr() if self.b is not None else None, _stream(dev)),
                       'sr_conv3x3_pack_f32')


def _conv_desc_f32(src, pc, out=None, *, upsample=False, act_slope=1.0, alpha=1.0, res1=None, beta1=0.0, res2=None,
                   beta2=0.0, accumulate=False, mask=None, mask_cb0=0, mask_slope=0.2, out_nchw=None):
    assert src.channels == pc.src_channels, (src.channels, pc.src_channels)
    H, W = (2 * src.h, 2 * src.w) if upsample else (src.h, src.w)
    d = _lib.ConvDesc()
    d.in_, d.in_img_stride, d.cin_pad, d.in_h, d.in_w = src.ptr, src.img_stride, pc.src_channels, src.h, src.w
    d.upsample = int(upsample)
    d.wpacked, d.bpacked, d.cout = pc.w.data_ptr(), (pc.b.data_ptr() if pc.b is not None else None), pc.cout
    if out_nchw is not None:
        assert out_nchw.is_contiguous() and out_nchw.shape == (src.n, pc.cout, H, W)
        d.out, d.out_img_stride, d.out_nchw = out_nchw.data_ptr(), pc.cout * H * W, 1
        ret = out_nchw
    else:
        if out is None:
            out = CB8.empty(src.n, pc.cout, H, W, src.device)
        assert (out.n, out.h, out.w) == (src.n, H, W) and out.channels >= (pc.cout + 7) // 8 * 8
        d.out, d.out_img_stride, d.out_nchw = out.ptr, out.img_stride, 0
        ret = out
    d.n, d.act_slope, d.alpha = src.n, act_slope, alpha
    if res1 is not None:
        d.res1, d.res1_img_stride, d.beta1 = res1.ptr, res1.img_stride, beta1
    if res2 is not None:
        d.res2, d.res2_img_stride, d.beta2 = res2.ptr, res2.img_stride, beta2
    d.accumulate = int(accumulate)
    if mask is not None:
        d.mask_src, d.mask_img_stride, d.mask_cb0, d.mask_cbn, d.mask_slope = (mask.ptr, mask.img_stride, mask_cb0,
                                                                               mask.cbn, mask_slope)
    return d, ret


def conv3x3(src, pc, out=None, **kw):
    """out = alpha*lrelu(conv(src)+bias) + beta1*res1 + beta2*res2  (one sr_conv3x3_f32 launch).

    src: CB8 window of pc.src_channels channels.  out: CB8 window (allocated if None) or, with
    ``out_nchw`` = an NCHW tensor [N, cout<=4, H, W], a plain tensor.  Keywords: upsample, act_slope, alpha, res1/beta1,
    res2/beta2, accumulate, mask/mask_cb0/mask_slope, out_nchw."""
    lib = _lib.load()
    d, ret = _conv_desc_f32(src, pc, out, **kw)
    with torch.cuda.device(src.device):
        _lib.check(lib.sr_conv3x3_f32(C.byref(d), _stream(src.device)), 'sr_conv3x3_f32')
    return ret


def conv3x3_chain(steps, sync=None, call_index=0):
    """fp32 twin of conv3x3_chain_bf16 — sr_conv3x3_chain_f32.  ``steps`` = [(src CB8, PackedConv, out CB8, kwargs of conv3x3)]."""
    lib = _lib.load()
    src0 = steps[0][0]
    if sync is None:
        sync = torch.zeros(lib.sr_conv3x3_chain_sync_ints(src0.n, src0.h, src0.w), dtype=torch.int32, device=src0.device)
    descs = (_lib.ConvDesc * len(steps))()
    outs = []
    for i, (src, pc, out, kw) in enumerate(steps):
        descs[i], ret = _conv_desc_f32(src, pc, out, **kw)
        outs.append(ret)
    with torch.cuda.device(src0.device):
        _lib.check(lib.sr_conv3x3_chain_f32(descs, len(steps), sync.data_ptr(), call_index, _stream(src0.device)), 'sr_conv3x3_chain_f32')
    return outs, sync


def conv3x3_wgrad(src, dy, cout, cin, first_seg=None, seg=0, *, upsample=False, scale=1.0, want_bias=True):
    """(dweight [cout,cin,3,3], dbias [cout]) of a 3x3 conv whose source was ``src`` (CB8) and whose
    pre-activation output gradient is ``dy`` (CB8) — one sr_conv3x3_wgrad_f32 call."""
    lib = _lib.load()
    first_seg = cin if first_seg is None else first_seg
    cin_pad = lib.sr_conv3x3_cin_pad(cin, first_seg, seg)
    assert src.channels == cin_pad, (src.channels, cin_pad)
    H, W = (2 * src.h, 2 * src.w) if upsample else (src.h, src.w)
    assert (dy.n, dy.h, dy.w) == (src.n, H, W) and dy.channels >= (cout + 7) // 8 * 8
    dev = src.device
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dev)
    db = torch.empty((cout,), dtype=torch.float32, device=dev) if want_bias else None
    nbytes = lib.sr_conv3x3_wgrad_slab_bytes(src.n, H, W)
    slab = scratch(dev, nbytes, 'slab')
    d = _lib.WgradDesc()
    d.x, d.x_img_stride, d.cin_pad, d.in_h, d.in_w, d.upsample = src.ptr, src.img_stride, cin_pad, src.h, src.w, int(upsample)
    d.dy, d.dy_img_stride = dy.ptr, dy.img_stride
    d.cout, d.cin, d.first_seg, d.seg, d.n, d.scale = cout, cin, first_seg, seg, src.n, scale
    d.dweight, d.dbias, d.accumulate = dw.data_ptr(), (db.data_ptr() if db is not None else None), 0
    d.slab, d.slab_bytes = slab.data_ptr(), nbytes
    with torch.cuda.device(dev):
        _lib.check(lib.sr_conv3x3_wgrad_f32(C.byref(d), _stream(dev)), 'sr_conv3x3_wgrad_f32')
    return dw, db


def upsample2x_bwd(g, mask=None, mask_slope=0.2):
    """2x2-sum backward of the nearest upsample (+ optional LeakyReLU backward) — sr_upsample2x_bwd_f32."""
    lib = _lib.load()
    assert g.h % 2 == 0 and g.w % 2 == 0
    out = CB8.empty(g.n, g.channels, g.h // 2, g.w // 2, g.device)
    with torch.cuda.device(g.device):
        _lib.check(lib.sr_upsample2x_bwd_f32(g.ptr, g.img_stride, out.ptr, out.img_stride,
                                             mask.ptr if mask is not None else None,
                                             mask.img_stride if mask is not None else 0, mask_slope, g.n, g.cbn,
                                             out.h, out.w, _stream(g.device)), 'sr_upsample2x_bwd_f32')
    return out


def cb8_axpby(dst, src, a=1.0, b=1.0):
    """dst = a*dst + b*src on CB8 windows — sr_cb8_axpby_f32."""
    lib = _lib.load()
    assert (dst.n, dst.cbn, dst.h, dst.w) == (src.n, src.cbn, src.h, src.w)
    with torch.cuda.device(dst.device):
        _lib.check(lib.sr_cb8_axpby_f32(dst.ptr, dst.img_stride, src.ptr, src.img_stride, a, b, dst.n, dst.cbn, dst.h,
                                        dst.w, _stream(dst.device)), 'sr_cb8_axpby_f32')
    return dst


class PackedConv4x4s2:
    """Parity-pass weight images of a 4x4 / stride 2 / pad 1 conv (sr_conv4x4s2_pack_f32); mode 1 = data gradient."""

    def __init__(self, weight, bias=None, mode=0):
        _need_cuda(weight, 'PackedConv4x4s2')
        lib = _lib.load()
        weight = weight.detach().contiguous().float()
        cout, cin = weight.shape[:2]
        assert weight.shape[2:] == (4, 4)
        self.mode, self.conv_cout, self.conv_cin = mode, cout, cin
        cin_pad = (cin + 7) // 8 * 8
        if mode == 0:
            self.cout, self.src_channels = cout, cin_pad
        else:
            self.cout, self.src_channels = cin_pad, (cout + 7) // 8 * 8
        dev = weight.device
        self.w = torch.empty(lib.sr_conv4x4s2_packed_weight_floats(cout, cin, mode), dtype=torch.float32, device=dev)
        self.b = None
        if mode == 0 and bias is not None:
            bias = bias.detach().contiguous().float()
            self.b = torch.empty(lib.sr_conv3x3_packed_bias_floats(cout), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_conv4x4s2_pack_f32(weight.data_ptr(), bias.data_ptr() if self.b is not None else None,
                                                 cout, cin, mode, self.w.data_ptr(),
                                                 self.b.data_ptr() if self.b is not None else None, _stream(dev)),
                       'sr_conv4x4s2_pack_f32')


def conv4x4s2(src, pc, out=None, *, act_slope=1.0, alpha=1.0):
    """Forward 4x4/s2/p1 conv (+bias, LeakyReLU): four accumulating parity passes of sr_conv4x4s2_f32."""
    lib = _lib.load()
    assert pc.mode == 0 and src.channels == pc.src_channels
    H, W = (src.h - 2) // 2 + 1, (src.w - 2) // 2 + 1
    if out is None:
        out = CB8.empty(src.n, pc.cout, H, W, src.device)
    d = _lib.ConvDesc()
    d.in_, d.in_img_stride, d.cin_pad, d.in_h, d.in_w = src.ptr, src.img_stride, pc.src_channels, src.h, src.w
    d.wpacked, d.bpacked, d.cout = pc.w.data_ptr(), (pc.b.data_ptr() if pc.b is not None else None), pc.cout
    d.out, d.out_img_stride, d.n, d.act_slope, d.alpha = out.ptr, out.img_stride, src.n, act_slope, alpha
    with torch.cuda.device(src.device):
        _lib.check(lib.sr_conv4x4s2_f32(C.byref(d), _stream(src.device)), 'sr_conv4x4s2_f32')
    return out


def conv4x4s2_dgrad(dy, pc, out_h, out_w, out=None, *, alpha=1.0, accumulate=False, mask=None, mask_cb0=0,
                    mask_slope=0.2):
    """dX of the 4x4/s2 conv from dY (sr_conv4x4s2_dgrad_f32)."""
    lib = _lib.load()
    assert pc.mode == 1 and dy.channels == pc.src_channels
    if out is None:
        out = CB8.empty(dy.n, pc.cout, out_h, out_w, dy.device)
    d = _lib.ConvDesc()
    d.in_, d.in_img_stride, d.cin_pad, d.in_h, d.in_w = dy.ptr, dy.img_stride, pc.src_channels, dy.h, dy.w
    d.wpacked, d.cout = pc.w.data_ptr(), pc.cout
    d.out, d.out_img_stride, d.out_h, d.out_w = out.ptr, out.img_stride, out_h, out_w
    d.n, d.act_slope, d.alpha, d.accumulate = dy.n, 1.0, alpha, int(accumulate)
    if mask is not None:
        d.mask_src, d.mask_img_stride, d.mask_cb0, d.mask_cbn, d.mask_slope = (mask.ptr, mask.img_stride, mask_cb0,
                                                                               mask.cbn, mask_slope)
    with torch.cuda.device(dy.device):
        _lib.check(lib.sr_conv4x4s2_dgrad_f32(C.byref(d), _stream(dy.device)), 'sr_conv4x4s2_dgrad_f32')
    return out


def conv4x4s2_wgrad(src, dy, cout, cin, *, scale=1.0, want_bias=False):
    """(dweight [cout,cin,4,4], dbias) of the 4x4/s2 conv (sr_conv4x4s2_wgrad_f32)."""
    lib = _lib.load()
    cin_pad = (cin + 7) // 8 * 8
    assert src.channels == cin_pad
    H, W = (src.h - 2) // 2 + 1, (src.w - 2) // 2 + 1
    assert (dy.n, dy.h, dy.w) == (src.n, H, W)
    dev = src.device
    dw = torch.zeros((cout, cin, 4, 4), dtype=torch.float32, device=dev)
    db = torch.empty((cout,), dtype=torch.float32, device=dev) if want_bias else None
    nbytes = lib.sr_conv3x3_wgrad_slab_bytes(src.n, H, W)
    slab = scratch(dev, nbytes, 'slab')
    d = _lib.WgradDesc()
    d.x, d.x_img_stride, d.cin_pad, d.in_h, d.in_w, d.upsample = src.ptr, src.img_stride, cin_pad, src.h, src.w, 0
    d.dy, d.dy_img_stride = dy.ptr, dy.img_stride
    d.cout, d.cin, d.first_seg, d.seg, d.n, d.scale = cout, cin, cin, 0, src.n, scale
    d.dweight, d.dbias, d.accumulate = dw.data_ptr(), (db.data_ptr() if db is not None else None), 0
    d.slab, d.slab_bytes = slab.data_ptr(), nbytes
    with torch.cuda.device(dev):
        _lib.check(lib.sr_conv4x4s2_wgrad_f32(C.byref(d), _stream(dev)), 'sr_conv4x4s2_wgrad_f32')
    return dw, db


# ------------------------------------------------------------------ bf16 (CB16) inference ops
class CB16:
    """A channel-blocked bf16 activation: storage ``buf`` [N, CB, H, W, 16] plus a channel-block window."""

    def __init__(self, buf, cb0=0, cbn=None):
        assert buf.dim() == 5 and buf.size(4) == 16 and buf.dtype == torch.bfloat16 and buf.is_contiguous()
        self.buf, self.cb0 = buf, cb0
        self.cbn = buf.size(1) - cb0 if cbn is None else cbn
        assert 0 <= cb0 and cb0 + self.cbn <= buf.size(1)

    @staticmethod
    def zeros(n, channels, h, w, device):
        return CB16(torch.zeros((n, (channels + 15) // 16, h, w, 16), dtype=torch.bfloat16, device=device))

    @staticmethod
    def empty(n, channels, h, w, device):
        return CB16(torch.empty((n, (channels + 15) // 16, h, w, 16), dtype=torch.bfloat16, device=device))

    n = property(lambda s: s.buf.size(0))
    h = property(lambda s: s.buf.size(2))
    w = property(lambda s: s.buf.size(3))
    channels = property(lambda s: s.cbn * 16)
    img_stride = property(lambda s: s.buf.size(1) * s.buf.size(2) * s.buf.size(3) * 16)
    device = property(lambda s: s.buf.device)

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.cb0 * self.h * self.w * 16 * 2

    def slice(self, c0, c):
        assert c0 % 16 == 0 and c % 16 == 0
        return CB16(self.buf, self.cb0 + c0 // 16, c // 16)


def nchw_to_cb16(x, unshuffle=1):
    """fp32 NCHW -> CB16 bf16 (round-to-nearest-even), pixel_unshuffle fused — sr_nchw_to_cb16_bf16."""
    lib = _lib.load()
    _need_cuda(x, 'nchw_to_cb16')
    x = x.contiguous().float()
    n, c, sh, sw = x.shape
    h, w = sh // unshuffle, sw // unshuffle
    out = CB16.empty(n, c * unshuffle * unshuffle, h, w, x.device)  # the kernel writes every block, pad channels as zero
    with torch.cuda.device(x.device):
        _lib.check(lib.sr_nchw_to_cb16_bf16(x.data_ptr(), out.ptr, n, c, h, w, unshuffle, out.cbn, out.img_stride,
                                            _stream(x.device)), 'sr_nchw_to_cb16_bf16')
    return out


def cb16_to_nchw(t, channels):
    lib = _lib.load()
    y = torch.empty((t.n, channels, t.h, t.w), dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        _lib.check(lib.sr_cb16_to_nchw_f32(t.ptr, t.img_stride, y.data_ptr(), t.n, channels, t.h, t.w, 1,
                                           _stream(t.device)), 'sr_cb16_to_nchw_f32')
    return y


class PackedConvBF16:
    """bf16 MFMA weight image (+ fp32 bias) of one 3x3 conv — sr_conv3x3_pack_bf16."""

    def __init__(self, weight, bias=None, first_seg=None, seg=0, mode=0):
        lib = _lib.load()
        _need_cuda(weight, 'PackedConvBF16')
        weight = weight.detach().contiguous().float()
        cout, cin = weight.shape[:2]
        first_seg = cin if first_seg is None else first_seg
        self.cin_pad = lib.sr_conv3x3_cin_pad16(cin, first_seg, seg)
        if self.cin_pad <= 0:
            raise ValueError(f'cin={cin} is not first_seg={first_seg} + k*seg={seg}')
        if mode == 0:
            self.cout, self.src_channels = cout, self.cin_pad
        else:  # data-gradient image: consumes dY (cout channels, padded to 16), produces the cin_pad source channels
            self.cout, self.src_channels = self.cin_pad, (cout + 15) // 16 * 16
        dev = weight.device
        self.w = torch.empty(lib.sr_conv3x3_packed_weight_elems_bf16(cout, cin, first_seg, seg, mode), dtype=torch.bfloat16,
                             device=dev)
        self.b = None
        if mode == 0 and bias is not None:
            bias = bias.detach().contiguous().float()
            self.b = torch.empty(lib.sr_conv3x3_packed_bias_floats(cout), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.sr_conv3x3_pack_bf16(weight.data_ptr(), bias.data_ptr() if self.b is not None else None, cout,
                                                cin, first_seg, seg, mode, self.w.data_ptr(),
                                                self.b.data_ptr() if self.b is not None else None, _stream(dev)),
                       'sr_conv3x3_pack_bf16')


def _conv_desc_bf16(src, pc, out=None, *, upsample=False, act_slope=1.0, alpha=1.0, res1=None, beta1=0.0, res2=None,
                    beta2=0.0, out_nchw=None, mask=None, mask_slope=0.2, s2_channels=0, s2_side=0, out_unshuffle2=False,
                    res1_u2=False, res1_keep_sign=False):
    assert src.channels == pc.src_channels, (src.channels, pc.src_channels)
    H, W = (2 * src.h, 2 * src.w) if upsample else (src.h, src.w)
    d = _lib.ConvDesc()
    d.in_, d.in_img_stride, d.cin_pad, d.in_h, d.in_w = src.ptr, src.img_stride, pc.src_channels, src.h, src.w
    d.upsample = int(upsample)
    d.wpacked, d.bpacked, d.cout = pc.w.data_ptr(), (pc.b.data_ptr() if pc.b is not None else None), pc.cout
    d.s2_channels, d.s2_side = s2_channels, s2_side
    if out_nchw is not None:
        assert out_nchw.is_contiguous() and out_nchw.dtype == torch.float32 and out_nchw.shape == (src.n, pc.cout, H, W)
        d.out, d.out_img_stride, d.out_nchw = out_nchw.data_ptr(), pc.cout * H * W, 1
        ret = out_nchw
    elif out_unshuffle2:
        # the destination only exists pixel-unshuffled: [n][4 cout / 16][H / 2][W / 2][16] (sr_conv3x3_desc.out_unshuffle2)
        assert out is None and pc.cout % 16 == 0 and H % 2 == 0 and W % 2 == 0
        out = CB16.empty(src.n, 4 * pc.cout, H // 2, W // 2, src.device)
        d.out, d.out_img_stride, d.out_nchw, d.out_unshuffle2 = out.ptr, out.img_stride, 0, 1
        ret = out
    else:
        if out is None:
            out = CB16.empty(src.n, pc.cout, H, W, src.device)  # every valid block is written (pad couts: zero weights)
        assert (out.n, out.h, out.w) == (src.n, H, W) and out.channels >= (pc.cout + 15) // 16 * 16
        d.out, d.out_img_stride, d.out_nchw = out.ptr, out.img_stride, 0
        ret = out
    d.n, d.act_slope, d.alpha = src.n, act_slope, alpha
    if res1 is not None:
        d.res1, d.res1_img_stride, d.beta1 = res1.ptr, res1.img_stride, beta1
        d.res1_u2, d.res1_keep_sign = int(res1_u2), int(res1_keep_sign)
    if res2 is not None:
        d.res2, d.res2_img_stride, d.beta2 = res2.ptr, res2.img_stride, beta2
    if mask is not None:
        d.mask_src, d.mask_img_stride, d.mask_cb0, d.mask_cbn, d.mask_slope = mask.ptr, mask.img_stride, 0, mask.cbn, mask_slope
    return d, ret


def conv3x3_bf16(src, pc, out=None, **kw):
    """bf16 twin of conv3x3 (fp32 accumulation and epilogue, bf16 CB16 or fp32 NCHW output) — sr_conv3x3_bf16.
    Keywords: upsample, act_slope, alpha, res1/beta1, res2/beta2, out_nchw, mask/mask_slope, s2_channels/s2_side
    (s2_channels = C marks a 4x4/s2 conv carried on a pixel-unshuffled operand of 4C channels: zero taps are skipped),
    out_unshuffle2 (the result is stored pixel-unshuffled only: [n][4 cout / 16][H / 2][W / 2][16])."""
    lib = _lib.load()
    d, ret = _conv_desc_bf16(src, pc, out, **kw)
    with torch.cuda.device(src.device):
        _lib.check(lib.sr_conv3x3_bf16(C.byref(d), _stream(src.device)), 'sr_conv3x3_bf16')
    return ret


def conv3x3_chain_bf16(steps, sync=None, call_index=0):
    """A dependency chain of convs as one persistent launch — sr_conv3x3_chain_bf16.  ``steps`` = [(src, pc, out, kwargs), ...] in
    execution order (conv k reads what earlier convs wrote); ``sync`` = int32 tensor of sr_conv3x3_chain_sync_ints(n, h, w) zeros
    (created when None).  Returns (outputs, sync); sync[0] != 0 after a synchronisation means a dependency wait timed out."""
    lib = _lib.load()
    src0 = steps[0][0]
    if sync is None:
        sync = torch.zeros(lib.sr_conv3x3_chain_sync_ints(src0.n, src0.h, src0.w), dtype=torch.int32, device=src0.device)
    descs = (_lib.ConvDesc * len(steps))()
    outs = []
    for i, (src, pc, out, kw) in enumerate(steps):
        descs[i], ret = _conv_desc_bf16(src, pc, out, **kw)
        outs.append(ret)
    with torch.cuda.device(src0.device):
        _lib.check(lib.sr_conv3x3_chain_bf16(descs, len(steps), sync.data_ptr(), call_index, _stream(src0.device)), 'sr_conv3x3_chain_bf16')
    return outs, sync


def conv3x3_wgrad_bf16(src, dy, cout, cin, first_seg=None, seg=0, *, upsample=False, scale=1.0, want_bias=True):
    """(dweight, dbias) in fp32 from bf16 CB16 source and output gradient — one sr_conv3x3_wgrad_bf16 call."""
    lib = _lib.load()
    first_seg = cin if first_seg is None else first_seg
    cin_pad = lib.sr_conv3x3_cin_pad16(cin, first_seg, seg)
    assert src.channels == cin_pad, (src.channels, cin_pad)
    H, W = (2 * src.h, 2 * src.w) if upsample else (src.h, src.w)
    assert (dy.n, dy.h, dy.w) == (src.n, H, W) and dy.channels >= (cout + 15) // 16 * 16
    dev = src.device
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=dev)
    db = torch.empty((cout,), dtype=torch.float32, device=dev) if want_bias else None
    nbytes = lib.sr_conv3x3_wgrad_slab_bytes_bf16(src.n, H, W)
    slab = scratch(dev, nbytes, 'slab16')
    d = _lib.WgradDesc()
    d.x, d.x_img_stride, d.cin_pad, d.in_h, d.in_w, d.upsample = src.ptr, src.img_stride, cin_pad, src.h, src.w, int(upsample)
    d.dy, d.dy_img_stride = dy.ptr, dy.img_stride
    d.cout, d.cin, d.first_seg, d.seg, d.n, d.scale = cout, cin, first_seg, seg, src.n, scale
    d.dweight, d.dbias, d.accumulate = dw.data_ptr(), (db.data_ptr() if db is not None else None), 0
    d.slab, d.slab_bytes = slab.data_ptr(), nbytes
    with torch.cuda.device(dev):
        _lib.check(lib.sr_conv3x3_wgrad_bf16(C.byref(d), _stream(dev)), 'sr_conv3x3_wgrad_bf16')
    return dw, db

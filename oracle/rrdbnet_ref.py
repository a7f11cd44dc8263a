"""Oracle: plain PyTorch-CPU fp32 restatement of the reference generator.

TEST INFRASTRUCTURE — not shipped, not measured as the product.  Pinned against the
reference itself: tools/make_goldens.py runs the reference's own modules
(/root/reference/Car_Plate-Restoration/basicsr/archs/rrdbnet_arch.py, imported in place)
on seeded inputs and commits the outputs under tests/golden/; tests/test_oracle.py checks
this file against those vectors.  Every function cites the reference lines it follows.

Parameters are passed as a flat dict name -> tensor with the reference's state_dict keys.
"""
import torch
import torch.nn.functional as F


def _t(v):
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def conv3x3(x, sd, name):
    """nn.Conv2d(cin, cout, 3, 1, 1) (rrdbnet_arch.py:21-25, 94-101)."""
    return F.conv2d(x, _t(sd[f'{name}.weight']), _t(sd[f'{name}.bias']), stride=1, padding=1)


def lrelu(x):
    """nn.LeakyReLU(negative_slope=0.2) (rrdbnet_arch.py:27, 103)."""
    return F.leaky_relu(x, 0.2)


def rdb_forward(x, sd, prefix='', return_intermediates=False):
    """ResidualDenseBlock.forward (rrdbnet_arch.py:32-39)."""
    x1 = lrelu(conv3x3(x, sd, f'{prefix}conv1'))
    x2 = lrelu(conv3x3(torch.cat((x, x1), 1), sd, f'{prefix}conv2'))
    x3 = lrelu(conv3x3(torch.cat((x, x1, x2), 1), sd, f'{prefix}conv3'))
    x4 = lrelu(conv3x3(torch.cat((x, x1, x2, x3), 1), sd, f'{prefix}conv4'))
    x5 = conv3x3(torch.cat((x, x1, x2, x3, x4), 1), sd, f'{prefix}conv5')
    out = x5 * 0.2 + x
    if return_intermediates:
        return out, (x1, x2, x3, x4)
    return out


def rrdb_forward(x, sd, prefix=''):
    """RRDB.forward (rrdbnet_arch.py:58-63)."""
    out = rdb_forward(x, sd, f'{prefix}rdb1.')
    out = rdb_forward(out, sd, f'{prefix}rdb2.')
    out = rdb_forward(out, sd, f'{prefix}rdb3.')
    return out * 0.2 + x


def pixel_unshuffle(x, scale):
    """arch_util.py:185-201."""
    b, c, hh, hw = x.size()
    assert hh % scale == 0 and hw % scale == 0
    h, w = hh // scale, hw // scale
    return x.view(b, c, h, scale, w, scale).permute(0, 1, 3, 5, 2, 4).reshape(b, c * scale * scale, h, w)


def head_forward(feat, sd):
    """Upsampling head of RRDBNet.forward (rrdbnet_arch.py:116-118)."""
    feat = lrelu(conv3x3(F.interpolate(feat, scale_factor=2, mode='nearest'), sd, 'conv_up1'))
    feat = lrelu(conv3x3(F.interpolate(feat, scale_factor=2, mode='nearest'), sd, 'conv_up2'))
    return conv3x3(lrelu(conv3x3(feat, sd, 'conv_hr')), sd, 'conv_last')


def rrdbnet_forward(x, sd, scale=4, num_block=23):
    """RRDBNet.forward (rrdbnet_arch.py:105-119)."""
    x = _t(x)
    if scale == 2:
        feat = pixel_unshuffle(x, 2)
    elif scale == 1:
        feat = pixel_unshuffle(x, 4)
    else:
        feat = x
    feat = conv3x3(feat, sd, 'conv_first')
    body = feat
    for b in range(num_block):
        body = rrdb_forward(body, sd, f'body.{b}.')
    feat = feat + conv3x3(body, sd, 'conv_body')
    return head_forward(feat, sd)

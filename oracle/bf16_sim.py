"""TEST INFRASTRUCTURE ONLY (imported by tests/ and tools/ checks, never by the product path).

Float64 model of what bf16 STORAGE does to the RRDBNet forward / backward of oracle/rrdbnet_ref.py (rrdbnet_arch.py:9-119):
the same network in float64 with a bf16 round trip wherever the HIP bf16 path stores a tensor — the input, every packed
weight, every conv output after its fused epilogue, and (through the backward of the same node) every activation gradient.
Accumulation stays exact, so a comparison against it isolates the kernels from the quantisation noise that any bf16
implementation has (parity unpinned: the reference has no reduced precision)."""
import torch
import torch.nn.functional as F

from .rrdbnet_ref import pixel_unshuffle


class _RoundTrip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(torch.float64)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(torch.float64)


def rrdbnet_forward_bf16_storage(x, sd, scale=4, num_block=23):
    """x float64 [N, C, H, W]; sd: float64 tensors with the reference's state_dict keys (requires_grad as wanted)."""
    r = _RoundTrip.apply

    def cv(t, name):
        w = sd[name + '.weight']
        wq = w.detach().to(torch.bfloat16).to(torch.float64) + (w - w.detach())  # bf16 value, gradient passed straight through
        return F.conv2d(t, wq, sd[name + '.bias'], padding=1)

    def lr(t):
        return F.leaky_relu(t, 0.2)
    feat = r(pixel_unshuffle(x, {4: 1, 2: 2, 1: 4}[scale]) if scale != 4 else x)
    feat = r(cv(feat, 'conv_first'))
    first = feat
    for b in range(num_block):
        x_rrdb = feat
        for k in (1, 2, 3):
            p, t = f'body.{b}.rdb{k}', feat
            x1 = r(lr(cv(t, p + '.conv1')))
            x2 = r(lr(cv(torch.cat((t, x1), 1), p + '.conv2')))
            x3 = r(lr(cv(torch.cat((t, x1, x2), 1), p + '.conv3')))
            x4 = r(lr(cv(torch.cat((t, x1, x2, x3), 1), p + '.conv4')))
            x5 = cv(torch.cat((t, x1, x2, x3, x4), 1), p + '.conv5')
            feat = r(x5 * 0.2 + t) if k < 3 else r((x5 * 0.2 + t) * 0.2 + x_rrdb)  # RRDB residual folded into rdb3.conv5's epilogue
    feat = r(first + cv(feat, 'conv_body'))
    feat = r(lr(cv(F.interpolate(feat, scale_factor=2, mode='nearest'), 'conv_up1')))
    feat = r(lr(cv(F.interpolate(feat, scale_factor=2, mode='nearest'), 'conv_up2')))
    return cv(r(lr(cv(feat, 'conv_hr'))), 'conv_last')


def unet_forward_bf16_storage(x, w, skip_connection=True):
    """UNetDiscriminatorSN (oracle/unet_discriminator_ref.py) in float64 with the bf16 round trips of the HIP bf16 path.
    `w`: effective (already spectrally normalised) weights 'conv0.weight', 'conv0.bias', 'conv1.weight' ... 'conv9.bias'."""
    r = _RoundTrip.apply

    def cv(t, name, stride=1):
        wt = w[name + '.weight']
        wq = wt.detach().to(torch.bfloat16).to(torch.float64) + (wt - wt.detach())
        return F.conv2d(t, wq, w.get(name + '.bias'), stride=stride, padding=1)

    def lr(t):
        return F.leaky_relu(t, 0.2)

    def up(t):
        return r(F.interpolate(t, scale_factor=2, mode='bilinear', align_corners=False))
    x0 = r(lr(cv(r(x), 'conv0')))
    x1 = r(lr(cv(x0, 'conv1', 2)))
    x2 = r(lr(cv(x1, 'conv2', 2)))
    x3 = r(lr(cv(x2, 'conv3', 2)))
    x4 = r(lr(cv(up(x3), 'conv4')))
    x5 = r(lr(cv(up(r(x4 + x2) if skip_connection else x4), 'conv5')))
    # (round 3: conv6 adds x0 in its epilogue — the activation itself is never stored, the sum is rounded once)
    x6 = lr(cv(up(r(x5 + x1) if skip_connection else x5), 'conv6'))
    x6 = r(x6 + x0) if skip_connection else r(x6)
    out = r(lr(cv(x6, 'conv7')))
    out = r(lr(cv(out, 'conv8')))
    return cv(out, 'conv9')

"""Oracle: PyTorch-CPU restatement of the reference's VGGStyleDiscriminator128 and of the ESRGAN losses.

TEST INFRASTRUCTURE ONLY.  Pinned against the reference itself through tests/golden/g_g_vgg128.npz and
g_h_losses.npz (tools/make_goldens.py); dtype-agnostic so tests can also run it in float64 to measure
how far fp32 summation order moves a gradient."""
import torch
import torch.nn.functional as F


def _t(v, like):
    v = v if isinstance(v, torch.Tensor) else torch.from_numpy(v)
    return v.to(like.dtype) if v.is_floating_point() else v


def _bn(x, sd, name, train, momentum=0.1, eps=1e-5):
    """nn.BatchNorm2d(affine=True) (discriminator_arch.py:23); running statistics are updated in place in train mode."""
    rm, rv = sd[f'{name}.running_mean'], sd[f'{name}.running_var']
    return F.batch_norm(x, rm, rv, sd[f'{name}.weight'], sd[f'{name}.bias'], train, momentum, eps)


def vgg128_forward(x, sd, train=True, input_size=128):
    """VGGStyleDiscriminator128.forward (discriminator_arch.py:51-72) and, with input_size=256, VGGStyleDiscriminator256.forward
    (:122-143: one more stage).  `sd`: dict of tensors with the reference's state_dict keys (running statistics are modified in
    place when train=True)."""
    assert x.size(2) == input_size and x.size(3) == input_size
    lrelu = lambda t: F.leaky_relu(t, 0.2)
    feat = lrelu(F.conv2d(x, sd['conv0_0.weight'], sd['conv0_0.bias'], 1, 1))
    feat = lrelu(_bn(F.conv2d(feat, sd['conv0_1.weight'], None, 2, 1), sd, 'bn0_1', train))
    for i in range(1, 6 if input_size == 256 else 5):
        feat = lrelu(_bn(F.conv2d(feat, sd[f'conv{i}_0.weight'], None, 1, 1), sd, f'bn{i}_0', train))
        feat = lrelu(_bn(F.conv2d(feat, sd[f'conv{i}_1.weight'], None, 2, 1), sd, f'bn{i}_1', train))
    feat = feat.view(feat.size(0), -1)
    feat = lrelu(F.linear(feat, sd['linear1.weight'], sd['linear1.bias']))
    return F.linear(feat, sd['linear2.weight'], sd['linear2.bias'])


def l1_loss(pred, target, loss_weight=1.0):
    """L1Loss(reduction='mean') (losses.py:98-106, l1_loss :65-67, weight_reduce_loss loss_util.py:25-54)."""
    return loss_weight * F.l1_loss(pred, target, reduction='mean')


def gan_loss(inp, target_is_real, is_disc, loss_weight=1.0):
    """GANLoss('vanilla').forward (losses.py:438-461): BCEWithLogits against ones/zeros; weight only for generators."""
    target = inp.new_ones(inp.size()) * (1.0 if target_is_real else 0.0)
    loss = F.binary_cross_entropy_with_logits(inp, target)
    return loss if is_disc else loss * loss_weight

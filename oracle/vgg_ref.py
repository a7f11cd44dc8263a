"""TEST INFRASTRUCTURE ONLY (oracle): PyTorch-CPU restatement of the VGG feature extractor of the perceptual loss
(reference basicsr/archs/vgg_arch.py:55-162; the convolution stack is torchvision's VGG ``features`` — configurations A, B,
D, E of arXiv:1409.1556 — which is not installed here).  **Parity unpinned**: neither torchvision nor its ImageNet weights are
available offline, so this restates the published architecture and is compared with the HIP path on random weights.
Imported by tests/ only."""
import torch
import torch.nn.functional as F

STAGES = {'vgg11': (1, 1, 2, 2, 2), 'vgg13': (2, 2, 2, 2, 2), 'vgg16': (2, 2, 3, 3, 3), 'vgg19': (2, 2, 4, 4, 4)}


def vgg_features(x, sd, layer_name_list, vgg_type='vgg19', use_input_norm=True, range_norm=False, remove_pooling=False):
    """sd: {'vgg_net.convS_K.weight' / '.bias': tensor}.  Returns {name: feature} (vgg_arch.py:143-161)."""
    if range_norm:
        x = (x + 1) / 2
    if use_input_norm:
        mean = torch.tensor([0.485, 0.456, 0.406], dtype=x.dtype).view(1, 3, 1, 1)
        std = torch.tensor([0.229, 0.224, 0.225], dtype=x.dtype).view(1, 3, 1, 1)
        x = (x - mean) / std
    out = {}
    want = set(layer_name_list)
    for stage, nconv in enumerate(STAGES[vgg_type], start=1):
        for k in range(1, nconv + 1):
            name = f'conv{stage}_{k}'
            if f'vgg_net.{name}.weight' not in sd:
                return out
            x = F.conv2d(x, sd[f'vgg_net.{name}.weight'].to(x.dtype), sd[f'vgg_net.{name}.bias'].to(x.dtype), padding=1)
            if name in want:
                out[name] = x
            x = F.relu(x)
            if f'relu{stage}_{k}' in want:
                out[f'relu{stage}_{k}'] = x
        if not remove_pooling:
            x = F.max_pool2d(x, 2, 2)
            if f'pool{stage}' in want:
                out[f'pool{stage}'] = x
    return out


def _gram(x):
    n, c, h, w = x.shape
    f = x.reshape(n, c, h * w)
    return f.bmm(f.transpose(1, 2)) / (c * h * w)


def perceptual_loss(x, gt, sd, layer_weights, perceptual_weight=1.0, style_weight=0.0, criterion='l1', **kw):
    """PerceptualLoss.forward (losses.py:301-356): returns the perceptual term, or (perceptual, style) when
    style_weight > 0; criterion 'l1' or 'fro'."""
    fx = vgg_features(x, sd, list(layer_weights), **kw)
    fg = vgg_features(gt.detach(), sd, list(layer_weights), **kw)

    def dist(a, b):
        return F.l1_loss(a, b) if criterion == 'l1' else torch.norm(a - b, p='fro')
    percep = sum(dist(fx[k], fg[k]) * w for k, w in layer_weights.items()) * perceptual_weight
    if style_weight > 0:
        return percep, sum(dist(_gram(fx[k]), _gram(fg[k])) * w for k, w in layer_weights.items()) * style_weight
    return percep

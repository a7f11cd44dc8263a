"""Oracle for UNetDiscriminatorSN — **parity unpinned by the reference** (SURVEY.md §0 D2, §8 a7).

TEST INFRASTRUCTURE ONLY.  The mounted reference does not contain this discriminator, so there are no reference
outputs to pin against; this is a PyTorch-CPU restatement of the published architecture (Real-ESRGAN,
arXiv:2107.10833 §3.3) using torch.nn.utils.spectral_norm itself, against which the HIP path is compared."""
import torch
from torch import nn
from torch.nn import functional as F
from torch.nn.utils import spectral_norm


class UNetDiscriminatorSNRef(nn.Module):

    def __init__(self, num_in_ch, num_feat=64, skip_connection=True):
        super().__init__()
        self.skip_connection = skip_connection
        norm = spectral_norm
        nf = num_feat
        self.conv0 = nn.Conv2d(num_in_ch, nf, 3, 1, 1)
        self.conv1 = norm(nn.Conv2d(nf, nf * 2, 4, 2, 1, bias=False))
        self.conv2 = norm(nn.Conv2d(nf * 2, nf * 4, 4, 2, 1, bias=False))
        self.conv3 = norm(nn.Conv2d(nf * 4, nf * 8, 4, 2, 1, bias=False))
        self.conv4 = norm(nn.Conv2d(nf * 8, nf * 4, 3, 1, 1, bias=False))
        self.conv5 = norm(nn.Conv2d(nf * 4, nf * 2, 3, 1, 1, bias=False))
        self.conv6 = norm(nn.Conv2d(nf * 2, nf, 3, 1, 1, bias=False))
        self.conv7 = norm(nn.Conv2d(nf, nf, 3, 1, 1, bias=False))
        self.conv8 = norm(nn.Conv2d(nf, nf, 3, 1, 1, bias=False))
        self.conv9 = nn.Conv2d(nf, 1, 3, 1, 1)

    def forward(self, x):
        x0 = F.leaky_relu(self.conv0(x), 0.2)
        x1 = F.leaky_relu(self.conv1(x0), 0.2)
        x2 = F.leaky_relu(self.conv2(x1), 0.2)
        x3 = F.leaky_relu(self.conv3(x2), 0.2)
        x3 = F.interpolate(x3, scale_factor=2, mode='bilinear', align_corners=False)
        x4 = F.leaky_relu(self.conv4(x3), 0.2)
        if self.skip_connection:
            x4 = x4 + x2
        x4 = F.interpolate(x4, scale_factor=2, mode='bilinear', align_corners=False)
        x5 = F.leaky_relu(self.conv5(x4), 0.2)
        if self.skip_connection:
            x5 = x5 + x1
        x5 = F.interpolate(x5, scale_factor=2, mode='bilinear', align_corners=False)
        x6 = F.leaky_relu(self.conv6(x5), 0.2)
        if self.skip_connection:
            x6 = x6 + x0
        out = F.leaky_relu(self.conv7(x6), 0.2)
        out = F.leaky_relu(self.conv8(out), 0.2)
        return self.conv9(out)

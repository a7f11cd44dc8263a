"""PSNR as the reference computes it during validation (basicsr/metrics/psnr_ssim.py:8-46 called from
sr_model.py:135-184 on ``tensor2img`` outputs): uint8-rounded images, optional border crop, 20*log10(255/sqrt(mse)).

``calculate_psnr`` is the host (numpy) form with the reference's signature; ``psnr_device`` keeps the SR output on the
GPU: clamp, *255, round (half-to-even like np.round), crop and the squared-error sum are one HIP reduction
(sr_psnr_sse_f32), so validation needs no device->host image copy."""
import math

import numpy as np
import torch

from ..utils.registry import METRIC_REGISTRY


def _to_y(img):
    # BT.601 luma of a BGR float image in [0, 255] (metric_util.to_y_channel -> bgr2ycbcr(y_only=True)), with the reference's
    # number formats: float32 input in [0, 1], float64 dot product, result stored as float32 in [0, 1], scaled back by 255
    img = img.astype(np.float32) / 255.
    y = ((np.dot(img, [24.966, 128.553, 65.481]) + 16.0) / 255.).astype(np.float32)
    return y[..., None] * 255.


@METRIC_REGISTRY.register()
def calculate_psnr(img1, img2, crop_border, input_order='HWC', test_y_channel=False):
    assert img1.shape == img2.shape, f'Image shapes are differnet: {img1.shape}, {img2.shape}.'
    if input_order not in ['HWC', 'CHW']:
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')
    if input_order == 'CHW':
        img1, img2 = img1.transpose(1, 2, 0), img2.transpose(1, 2, 0)
    if img1.ndim == 2:
        img1, img2 = img1[..., None], img2[..., None]
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    if crop_border != 0:
        img1 = img1[crop_border:-crop_border, crop_border:-crop_border, ...]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        img1, img2 = _to_y(img1), _to_y(img2)
    mse = np.mean((img1 - img2)**2)
    if mse == 0:
        return float('inf')
    return 20. * np.log10(255. / np.sqrt(mse))


def psnr_device(sr, gt, crop_border=0):
    """PSNR per image of NCHW float tensors in [0, 1] on the HIP device, with tensor2img's quantisation."""
    import ctypes as C
    from .. import _lib
    from ..hip_ops import scratch
    assert sr.shape == gt.shape and sr.dim() == 4 and sr.is_cuda
    lib = _lib.load()
    sr, gt = sr.contiguous().float(), gt.contiguous().float()
    n, c, h, w = sr.shape
    out = torch.empty(n, dtype=torch.float32, device=sr.device)
    wsb = lib.sr_reduce_workspace_bytes(8) * max(n, 1)
    ws = scratch(sr.device, wsb)
    with torch.cuda.device(sr.device):
        _lib.check(lib.sr_psnr_sse_f32(sr.data_ptr(), gt.data_ptr(), n, c, h, w, crop_border, out.data_ptr(), ws.data_ptr(), wsb,
                                       torch.cuda.current_stream(sr.device).cuda_stream), 'sr_psnr_sse_f32')
    count = c * (h - 2 * crop_border) * (w - 2 * crop_border)
    mse = out.double().cpu().numpy() / count
    return [float('inf') if m == 0 else 20. * math.log10(255. / math.sqrt(m)) for m in mse]


def _gauss11():
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))  # cv2.getGaussianKernel(11, 1.5)
    return g / g.sum()


def _ssim(img1, img2):
    """One channel, float64, valid region of the separable 11x11 Gaussian statistics (psnr_ssim.py:49-83; the
    reference filters with cv2.filter2D and crops 5 pixels, which is the same valid region)."""
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    g = _gauss11()

    def filt(a):
        a = np.apply_along_axis(lambda r: np.convolve(r, g, mode='valid'), 1, a)
        return np.apply_along_axis(lambda r: np.convolve(r, g, mode='valid'), 0, a)
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    mu1, mu2 = filt(img1), filt(img2)
    s11, s22, s12 = filt(img1 ** 2) - mu1 ** 2, filt(img2 ** 2) - mu2 ** 2, filt(img1 * img2) - mu1 * mu2
    m = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 ** 2 + mu2 ** 2 + c1) * (s11 + s22 + c2))
    return m.mean()


@METRIC_REGISTRY.register()
def calculate_ssim(img1, img2, crop_border, input_order='HWC', test_y_channel=False):
    """SSIM of uint8-range images with the reference's signature (psnr_ssim.py:86-128): per channel, then averaged."""
    assert img1.shape == img2.shape, f'Image shapes are differnet: {img1.shape}, {img2.shape}.'
    if input_order not in ['HWC', 'CHW']:
        raise ValueError(f'Wrong input_order {input_order}. Supported input_orders are "HWC" and "CHW"')
    if input_order == 'CHW':
        img1, img2 = img1.transpose(1, 2, 0), img2.transpose(1, 2, 0)
    if img1.ndim == 2:
        img1, img2 = img1[..., None], img2[..., None]
    img1, img2 = img1.astype(np.float64), img2.astype(np.float64)
    if crop_border != 0:
        img1 = img1[crop_border:-crop_border, crop_border:-crop_border, ...]
        img2 = img2[crop_border:-crop_border, crop_border:-crop_border, ...]
    if test_y_channel:
        img1, img2 = _to_y(img1), _to_y(img2)
    return float(np.mean([_ssim(img1[..., i], img2[..., i]) for i in range(img1.shape[2])]))


def ssim_device(sr, gt, crop_border=0):
    """SSIM per image of NCHW float tensors in [0, 1] on the HIP device, with tensor2img's quantisation (channel order
    does not matter: the channel mean is symmetric)."""
    import ctypes as C
    from .. import _lib
    from ..hip_ops import scratch
    assert sr.shape == gt.shape and sr.dim() == 4 and sr.is_cuda
    lib = _lib.load()
    sr, gt = sr.contiguous().float(), gt.contiguous().float()
    n, c, h, w = sr.shape
    out = torch.empty(n, dtype=torch.float32, device=sr.device)
    wsb = lib.sr_reduce_workspace_bytes(8) * max(n, 1)
    ws = scratch(sr.device, wsb)
    with torch.cuda.device(sr.device):
        _lib.check(lib.sr_ssim_sum_f32(sr.data_ptr(), gt.data_ptr(), n, c, h, w, crop_border, out.data_ptr(), ws.data_ptr(), wsb,
                                       torch.cuda.current_stream(sr.device).cuda_stream), 'sr_ssim_sum_f32')
    count = c * (h - 2 * crop_border - 10) * (w - 2 * crop_border - 10)
    return [float(v) / count for v in out.double().cpu().numpy()]

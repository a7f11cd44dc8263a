"""Validation metrics of the path: PSNR and SSIM on uint8-rounded images, host (numpy) and device (HIP)."""
from .psnr import calculate_psnr, calculate_ssim, psnr_device, ssim_device  # noqa: F401

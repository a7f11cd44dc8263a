"""Validation metrics of the path: PSNR on uint8-rounded images, host (numpy) and device (HIP)."""
from .psnr import calculate_psnr, psnr_device  # noqa: F401

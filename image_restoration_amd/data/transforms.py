"""Paired crop and flip / rotate augmentation of the training pipeline.

Counterparts of basicsr/data/transforms.py:26-158 for HWC numpy images.  The random draws are made with Python's ``random``
in the reference's order (top, left; hflip, vflip, rot90), so a seeded run picks the same patches and symmetries.  The
reference flips through cv2 in place; here flips are numpy views made contiguous (cv2 is not installed)."""
import random

import numpy as np


def mod_crop(img, scale):
    """Crop H and W down to multiples of ``scale`` (transforms.py:6-23)."""
    img = img.copy()
    if img.ndim not in (2, 3):
        raise ValueError(f'Wrong img ndim: {img.ndim}.')
    h, w = img.shape[0], img.shape[1]
    return img[:h - h % scale, :w - w % scale, ...]


def paired_random_crop(img_gts, img_lqs, gt_patch_size, scale, gt_path=None):
    """Same random LQ window and the matching GT window (x scale) for every image of the two lists."""
    if not isinstance(img_gts, list):
        img_gts = [img_gts]
    if not isinstance(img_lqs, list):
        img_lqs = [img_lqs]
    h_lq, w_lq = img_lqs[0].shape[0:2]
    h_gt, w_gt = img_gts[0].shape[0:2]
    lq_patch_size = gt_patch_size // scale
    if h_gt != h_lq * scale or w_gt != w_lq * scale:
        raise ValueError(f'Scale mismatches. GT ({h_gt}, {w_gt}) is not {scale}x multiplication of LQ ({h_lq}, {w_lq}).')
    if h_lq < lq_patch_size or w_lq < lq_patch_size:
        raise ValueError(f'LQ ({h_lq}, {w_lq}) is smaller than patch size ({lq_patch_size}, {lq_patch_size}). '
                         f'Please remove {gt_path}.')
    top = random.randint(0, h_lq - lq_patch_size)
    left = random.randint(0, w_lq - lq_patch_size)
    img_lqs = [v[top:top + lq_patch_size, left:left + lq_patch_size, ...] for v in img_lqs]
    top_gt, left_gt = int(top * scale), int(left * scale)
    img_gts = [v[top_gt:top_gt + gt_patch_size, left_gt:left_gt + gt_patch_size, ...] for v in img_gts]
    return (img_gts[0] if len(img_gts) == 1 else img_gts), (img_lqs[0] if len(img_lqs) == 1 else img_lqs)


def augment(imgs, hflip=True, rotation=True, return_status=False):
    """Horizontal flip, vertical flip and transpose, each with probability 1/2, identical for all images."""
    hflip = hflip and random.random() < 0.5
    vflip = rotation and random.random() < 0.5
    rot90 = rotation and random.random() < 0.5

    def one(img):
        if hflip:
            img = img[:, ::-1, ...]
        if vflip:
            img = img[::-1, :, ...]
        if rot90:
            img = img.transpose(1, 0, 2) if img.ndim == 3 else img.transpose(1, 0)
        return np.ascontiguousarray(img)

    single = not isinstance(imgs, list)
    out = [one(img) for img in ([imgs] if single else imgs)]
    out = out[0] if len(out) == 1 else out
    return (out, (hflip, vflip, rot90)) if return_status else out

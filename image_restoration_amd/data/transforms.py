"""Paired patch geometry of the training pipeline: window draw, symmetry draw, and their application.

Behavioural counterparts of basicsr/data/transforms.py:6-158 for HWC numpy images: ``mod_crop``, ``paired_random_crop``,
``augment`` keep the reference's signatures and — what a seeded run can observe — its consumption of Python's ``random``:
``randint`` for top then left, then one ``random()`` each for horizontal flip, vertical flip and transposition, in that order
(golden G-p pins the windows against the reference itself).

Own structure: a draw and its application are separate steps.  ``draw_window`` / ``draw_symmetry`` return plain integers, so the
SAME decisions can be applied on the host (``cut_pair`` / ``apply_symmetry`` below: numpy views) or on the device
(``device_pipeline.DevicePatchPipeline``: one HIP kernel over a whole uint8 batch in HBM, sr_patch_augment_u8_f32).
The reference flips with cv2 in place; cv2 is not installed and not needed: flips are index reversals.
"""
import random

import numpy as np

SYM_HFLIP, SYM_VFLIP, SYM_TRANSPOSE = 1, 2, 4   # bit flags of a symmetry code, also the kernel's


def mod_crop(img, scale):
    """The largest top-left crop whose height and width are multiples of ``scale`` (a copy)."""
    if img.ndim != 2 and img.ndim != 3:
        raise ValueError(f'Wrong img ndim: {img.ndim}.')
    rows, cols = img.shape[:2]
    return img[:rows - rows % scale, :cols - cols % scale].copy()


def draw_window(lq_rows, lq_cols, lq_patch):
    """(top, left) of an ``lq_patch`` window inside an ``lq_rows x lq_cols`` image: two ``random.randint`` draws."""
    return random.randint(0, lq_rows - lq_patch), random.randint(0, lq_cols - lq_patch)


def draw_symmetry(hflip=True, rotation=True):
    """Symmetry code (bit flags above), three fair coins: ``random.random() < 0.5`` is only evaluated for enabled symmetries,
    like the reference's short-circuit ``and``."""
    code = 0
    if hflip and random.random() < 0.5:
        code |= SYM_HFLIP
    if rotation and random.random() < 0.5:
        code |= SYM_VFLIP
    if rotation and random.random() < 0.5:
        code |= SYM_TRANSPOSE
    return code


def apply_symmetry(img, code):
    """Reverse columns, reverse rows, swap the two image axes — in that order, as far as ``code`` says; contiguous result."""
    if code & SYM_HFLIP:
        img = img[:, ::-1]
    if code & SYM_VFLIP:
        img = img[::-1]
    if code & SYM_TRANSPOSE:
        img = np.swapaxes(img, 0, 1)
    return np.ascontiguousarray(img)


def _as_list(x):
    return (x, True) if isinstance(x, list) else ([x], False)


def check_pair_geometry(gt_shape, lq_shape, gt_patch_size, scale, gt_path=None):
    """The reference's two refusals: GT must be exactly ``scale`` x LQ, and the LQ must hold one patch."""
    (gh, gw), (lh, lw) = gt_shape[:2], lq_shape[:2]
    need = gt_patch_size // scale
    if (gh, gw) != (lh * scale, lw * scale):
        raise ValueError(f'Scale mismatches. GT ({gh}, {gw}) is not {scale}x multiplication of LQ ({lh}, {lw}).')
    if min(lh, lw) < need:
        raise ValueError(f'LQ ({lh}, {lw}) is smaller than patch size ({need}, {need}). Please remove {gt_path}.')
    return need


def cut_pair(img_gts, img_lqs, top, left, lq_patch, scale, gt_patch=None):
    gt_patch = lq_patch * scale if gt_patch is None else gt_patch
    gts = [g[top * scale:top * scale + gt_patch, left * scale:left * scale + gt_patch] for g in img_gts]
    lqs = [q[top:top + lq_patch, left:left + lq_patch] for q in img_lqs]
    return gts, lqs


def paired_random_crop(img_gts, img_lqs, gt_patch_size, scale, gt_path=None):
    """One random LQ window and the GT window it corresponds to, for an image pair or for two lists of images."""
    gts, many_gt = _as_list(img_gts)
    lqs, many_lq = _as_list(img_lqs)
    lq_patch = check_pair_geometry(gts[0].shape, lqs[0].shape, gt_patch_size, scale, gt_path)
    top, left = draw_window(lqs[0].shape[0], lqs[0].shape[1], lq_patch)
    gts, lqs = cut_pair(gts, lqs, top, left, lq_patch, scale, gt_patch_size)
    return (gts if many_gt and len(gts) > 1 else gts[0]), (lqs if many_lq and len(lqs) > 1 else lqs[0])


def augment(imgs, hflip=True, rotation=True, return_status=False):
    """The same random symmetry for every image given; with ``return_status`` also (hflip, vflip, rot90) as booleans."""
    code = draw_symmetry(hflip, rotation)
    batch, _ = _as_list(imgs)
    done = [apply_symmetry(im, code) for im in batch]
    result = done[0] if len(done) == 1 else done
    if return_status:
        return result, (bool(code & SYM_HFLIP), bool(code & SYM_VFLIP), bool(code & SYM_TRANSPOSE))
    return result

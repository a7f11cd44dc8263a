"""DevicePatchPipeline: the per-batch half of the training input pipeline on the MI355X.

``PairedImageDataset(device_augment: true)`` ships, per sample, the uint8 LQ / GT windows it cut (or, with ``device_augment: full``
on equally sized images, the whole uint8 images plus the window origin) and the symmetry code it drew.  This class turns the
collated batch — already in HBM, on the feed's copy stream — into the tensors the model trains on with one HIP launch per tensor
(sr_patch_augment_u8_f32): crop, hflip / vflip / transpose, BGR->RGB, HWC->CHW, /255, (x - mean) / std.  Same draws, same
arithmetic: the result is bit-identical to the host pipeline of the reference (basicsr/data/paired_image_dataset.py:67-98).
"""
import ctypes as C

import torch

from .. import _lib


def _f3(values):
    return None if values is None else (C.c_float * 3)(*[float(v) for v in values])


class DevicePatchPipeline:

    def __init__(self, dataset_opt):
        self.scale = int(dataset_opt.get('scale', 4))
        self.gt_size = int(dataset_opt['gt_size'])
        mean, std = dataset_opt.get('mean'), dataset_opt.get('std')
        if (mean is None) != (std is None):   # normalize() with one side missing: identity on the other, like ImageSource
            mean = mean if mean is not None else [0.0] * 3
            std = std if std is not None else [1.0] * 3
        self._mean, self._std = _f3(mean), _f3(std)

    def _launch(self, src, origin, origin_mul, sym, patch):
        if not src.is_cuda:
            raise _lib.SrHipError('DevicePatchPipeline runs only on a HIP device (no CPU fallback)')
        if src.dtype != torch.uint8 or src.dim() != 4 or src.size(3) != 3 or not src.is_contiguous():
            raise ValueError(f'expected a contiguous uint8 [B, H, W, 3] batch, got {src.dtype} {tuple(src.shape)}')
        b, h, w, _ = src.shape
        out = torch.empty(b, 3, patch, patch, dtype=torch.float32, device=src.device)
        top = left = None
        if origin is not None:
            top, left = origin[:, 0].contiguous(), origin[:, 1].contiguous()
        lib = _lib.load()
        with torch.cuda.device(src.device):
            _lib.check(lib.sr_patch_augment_u8_f32(
                src.data_ptr(), 0, h, w, top.data_ptr() if top is not None else None, left.data_ptr() if left is not None else None,
                origin_mul, sym.data_ptr() if sym is not None else None, out.data_ptr(), b, patch, patch, 1, self._mean, self._std,
                torch.cuda.current_stream(src.device).cuda_stream), 'sr_patch_augment_u8_f32')
        return out

    def __call__(self, batch):
        if 'lq_u8' not in batch:
            return batch   # a host-augmented batch passes through
        sym = batch['sym'].to(torch.int32).contiguous()
        origin = batch.get('window')          # [B, 2] (top, left) in LQ pixels when whole images were shipped
        if origin is not None:
            origin = origin.to(torch.int32)
        out = {k: v for k, v in batch.items() if k not in ('lq_u8', 'gt_u8', 'sym', 'window')}
        out['lq'] = self._launch(batch['lq_u8'], origin, 1, sym, self.gt_size // self.scale)
        out['gt'] = self._launch(batch['gt_u8'], origin, self.scale, sym, self.gt_size)
        return out

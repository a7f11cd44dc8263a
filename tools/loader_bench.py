"""Input-pipeline throughput: PairedImageDataset over a PNG folder -> DataLoader workers -> DeviceFeed, with the
augmentation on the host (the reference's pipeline: float crops, flips, HWC->CHW per sample in the workers) or on the device
(device_augment: uint8 windows over PCIe, one HIP launch per batch).  Prints images/s for both next to the model step they feed.

    python tools/loader_bench.py [--workers 8] [--gt 128] [--batch 32] [--images 64] [--batches 60]"""
import argparse
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from PIL import Image
from image_restoration_amd.data import DeviceFeed, DevicePatchPipeline, EnlargedSampler, PairedImageDataset


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workers', type=int, default=8)
    ap.add_argument('--gt', type=int, default=128, help='GT patch size (reference recipe: 128; BASELINE config 3: 512)')
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--images', type=int, default=64)
    ap.add_argument('--size', type=int, default=480, help='GT image side (DIV2K sub-images of the reference recipe: 480)')
    ap.add_argument('--batches', type=int, default=60)
    args = ap.parse_args()
    size = max(args.size, args.gt)
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as tmp:
        gt_dir, lq_dir = os.path.join(tmp, 'gt'), os.path.join(tmp, 'lq')
        os.makedirs(gt_dir), os.makedirs(lq_dir)
        for i in range(args.images):
            # smooth content + noise: PNGs that compress like photographs rather than like white noise
            base = rng.integers(0, 256, (size // 8, size // 8, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)
            gt = np.clip(base.astype(np.int16) + rng.integers(-6, 7, base.shape), 0, 255).astype(np.uint8)
            Image.fromarray(gt).save(os.path.join(gt_dir, f'{i:04d}.png'))
            Image.fromarray(gt[::4, ::4]).save(os.path.join(lq_dir, f'{i:04d}.png'))
        base = dict(name='bench', type='PairedImageDataset', dataroot_gt=gt_dir, dataroot_lq=lq_dir, filename_tmpl='{}',
                    io_backend=dict(type='disk'), scale=4, phase='train', gt_size=args.gt, use_flip=True, use_rot=True)
        for label, extra in (('host augmentation', {}), ('device augmentation', dict(device_augment=True))):
            ds = PairedImageDataset(dict(base, **extra))
            ratio = max(1, (args.batches + 8) * args.batch // len(ds) + 1)
            sampler = EnlargedSampler(ds, 1, 0, ratio)
            loader = torch.utils.data.DataLoader(ds, batch_size=args.batch, sampler=sampler, num_workers=args.workers, drop_last=True,
                                                 pin_memory=True, persistent_workers=args.workers > 0)
            feed = DeviceFeed(loader, dict(num_gpu=1), pipeline=DevicePatchPipeline(base) if extra else None)
            for _ in range(6):
                b = feed.next()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.batches):
                b = feed.next()
                assert b is not None and b['gt'].shape == (args.batch, 3, args.gt, args.gt) and b['gt'].is_cuda
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f'{label:20s}: {args.batches * args.batch / dt:9.1f} images/s  ({args.workers} workers, batch {args.batch}, '
                  f'{size}x{size} GT PNGs -> {args.gt}x{args.gt} patches)', flush=True)
            del feed, loader


if __name__ == '__main__':
    main()

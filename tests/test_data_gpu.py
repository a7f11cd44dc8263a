"""Device side of the input pipeline: sr_patch_augment_u8_f32 / DevicePatchPipeline against the host pipeline that
restates basicsr/data/{transforms.py:26-158, paired_image_dataset.py:67-98} — same random draws, bit-identical tensors — and
against the reference's own crop windows (golden G-p)."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

from image_restoration_amd.data import DeviceFeed, DevicePatchPipeline, PairedImageDataset
from image_restoration_amd.data import transforms as T

pytestmark = pytest.mark.gpu


def _pairs(tmp_path, sizes, seed=0):
    rng = np.random.default_rng(seed)
    gt_dir, lq_dir = tmp_path / 'gt', tmp_path / 'lq'
    gt_dir.mkdir(), lq_dir.mkdir()
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(rng.integers(0, 256, (h * 4, w * 4, 3), dtype=np.uint8)).save(gt_dir / f'im{i}.png')
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(lq_dir / f'im{i}x4.png')
    return dict(name='t', type='PairedImageDataset', dataroot_gt=str(gt_dir), dataroot_lq=str(lq_dir), filename_tmpl='{}x4',
                io_backend=dict(type='disk'), scale=4, phase='train', gt_size=64, use_flip=True, use_rot=True)


@pytest.mark.parametrize('mode,norm', [(True, False), ('full', False), (True, True), ('full', True)])
def test_device_pipeline_is_bit_identical_to_the_host_pipeline(cuda, tmp_path, mode, norm):
    """Random uint8 images (every byte value, so the /255 and normalisation roundings are all exercised), 24 samples covering all 8
    symmetry codes: crop + flips + transpose + BGR->RGB + CHW + /255 [+ (x-mean)/std] on the device == the host datasets' tensors,
    bit for bit, for host-cut windows (device_augment: true) and device-side cropping of whole images (device_augment: full)."""
    sizes = [(40, 52)] * 6 if mode == 'full' else [(40, 52), (16, 16), (33, 47), (64, 20), (21, 22), (18, 90)]
    base = _pairs(tmp_path, sizes)
    if norm:
        base.update(mean=[0.4, 0.5, 0.45], std=[0.25, 0.5, 0.2])
    host, dev = PairedImageDataset(dict(base)), PairedImageDataset(dict(base, device_augment=mode))
    pipe = DevicePatchPipeline(base)
    seen = set()
    for rep in range(10):
        want, items = [], []
        for i in range(len(sizes)):
            random.seed(1000 * rep + i)
            want.append(host[i])
            random.seed(1000 * rep + i)
            items.append(dev[i])
        batch = torch.utils.data.default_collate(items)
        seen |= set(int(s) for s in batch['sym'])
        out = pipe({k: (v.to(cuda) if torch.is_tensor(v) else v) for k, v in batch.items()})
        assert out['lq'].shape == (len(sizes), 3, 16, 16) and out['gt'].shape == (len(sizes), 3, 64, 64) and 'lq_u8' not in out
        assert out['gt_path'] == [w['gt_path'] for w in want]
        assert torch.equal(out['lq'].cpu(), torch.stack([w['lq'] for w in want]))
        assert torch.equal(out['gt'].cpu(), torch.stack([w['gt'] for w in want]))
    assert seen == set(range(8))


def test_device_crop_takes_the_reference_windows(cuda, golden):
    """Golden G-p holds the LQ windows (and GT window corners) the REFERENCE's paired_random_crop cut for seeds 0..15 out of a
    coordinate-coded 20x24 / 80x96 pair (tools/make_goldens.py g_p: channels = row, column, (row + column) % 7).  The device kernel,
    given whole uint8 images and the draws of transforms.draw_window under the same seeds, returns exactly those windows."""
    g = golden('g_p_crop')
    yy, xx = np.meshgrid(np.arange(20), np.arange(24), indexing='ij')
    lq_u8 = np.stack([yy, xx, (yy + xx) % 7], axis=2).astype(np.uint8)      # the golden's image, exactly representable in uint8
    gt_u8 = lq_u8.repeat(4, 0).repeat(4, 1)
    pipe = DevicePatchPipeline(dict(scale=4, gt_size=32))
    windows = []
    for seed in range(16):
        random.seed(seed)
        windows.append(T.draw_window(20, 24, 8))
    n = 16
    batch = {'lq_u8': torch.from_numpy(np.stack([lq_u8] * n)).to(cuda), 'gt_u8': torch.from_numpy(np.stack([gt_u8] * n)).to(cuda),
             'sym': torch.zeros(n, dtype=torch.int32, device=cuda), 'window': torch.tensor(windows, dtype=torch.int32, device=cuda)}
    out = pipe(batch)
    back = lambda t: (t.cpu().numpy() * 255.).round().astype(np.float32).transpose(1, 2, 0)[:, :, ::-1]   # CHW RGB /255 -> HWC BGR
    for seed in range(n):
        assert np.array_equal(back(out['lq'][seed]), g[f'lq_{seed}']), seed
        assert np.array_equal(back(out['gt'][seed])[::31, ::31], g[f'gt_corner_{seed}']), seed


def test_device_feed_stages_and_augments_on_the_copy_stream(cuda, tmp_path):
    """DeviceFeed + DevicePatchPipeline over a DataLoader (two epochs through reset()): float batches on the device, equal to the host
    pipeline's under the same per-item seeds; the staging buffers of batch k+1 do not disturb batch k while it is in use."""
    base = _pairs(tmp_path, [(24, 24)] * 5, seed=3)
    ds = PairedImageDataset(dict(base, device_augment=True))
    host = PairedImageDataset(dict(base))

    class Seeded(torch.utils.data.Dataset):
        def __init__(self, inner):
            self.inner = inner

        def __len__(self):
            return len(self.inner)

        def __getitem__(self, i):
            random.seed(77 + i)
            return self.inner[i]

    loader = torch.utils.data.DataLoader(Seeded(ds), batch_size=2, shuffle=False, num_workers=0, pin_memory=True)
    feed = DeviceFeed(loader, dict(num_gpu=1), pipeline=DevicePatchPipeline(base))
    want = [Seeded(host)[i] for i in range(5)]
    for _ in range(2):
        feed.reset()
        got, held = [], []
        while (b := feed.next()) is not None:
            assert b['lq'].is_cuda and b['lq'].dtype == torch.float32
            held.append(b)           # keep every batch alive while later ones are staged
            torch.cuda.current_stream().synchronize()
        for b in held:
            got += list(b['gt'].cpu())
        assert len(got) == 5 and all(torch.equal(a, w['gt']) for a, w in zip(got, want))


def test_patch_augment_rejects_bad_arguments(cuda):
    from image_restoration_amd import _lib
    pipe = DevicePatchPipeline(dict(scale=4, gt_size=64))
    with pytest.raises(ValueError):
        pipe({'lq_u8': torch.zeros(2, 16, 16, 3, device=cuda), 'gt_u8': torch.zeros(2, 64, 64, 3, dtype=torch.uint8, device=cuda),
              'sym': torch.zeros(2, dtype=torch.int32, device=cuda)})          # float instead of uint8
    with pytest.raises(_lib.SrHipError):
        pipe({'lq_u8': torch.zeros(2, 8, 8, 3, dtype=torch.uint8, device=cuda), 'gt_u8': torch.zeros(2, 64, 64, 3, dtype=torch.uint8, device=cuda),
              'sym': torch.zeros(2, dtype=torch.int32, device=cuda)})          # window larger than the source
    with pytest.raises(_lib.SrHipError):
        pipe({'lq_u8': torch.zeros(2, 16, 16, 3, dtype=torch.uint8), 'gt_u8': torch.zeros(2, 64, 64, 3, dtype=torch.uint8),
              'sym': torch.zeros(2, dtype=torch.int32)})                       # host tensors: no CPU fallback

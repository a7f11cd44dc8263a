"""Host input pipeline: path pairing, paired crop / augmentation geometry, PairedImageDataset and the
prefetchers, on temporary PNG folders.  cv2 / lmdb are not in the image, so this is pinned to the published behaviour of
basicsr/data/{paired_image_dataset,transforms,data_util}.py only (geometry and value-range properties, seeded draws)."""
import os
import random

import numpy as np
import pytest
import torch
from PIL import Image

from image_restoration_amd.data import CPUPrefetcher, CUDAPrefetcher, PairedImageDataset, SingleImageDataset
from image_restoration_amd.data.data_util import paired_paths_from_folder, paired_paths_from_meta_info_file
from image_restoration_amd.data.transforms import augment, mod_crop, paired_random_crop
from image_restoration_amd.utils.img_util import imfrombytes
from image_restoration_amd.utils.registry import DATASET_REGISTRY


def _coord_pair(h, w, scale):
    """LQ image whose pixel encodes (y, x); GT = its nearest x scale upsampling, so crops / flips stay checkable."""
    yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing='ij')
    lq = np.stack([yy, xx, (yy + xx) % 7], axis=2).astype(np.uint8)
    return lq, lq.repeat(scale, 0).repeat(scale, 1)


@pytest.fixture()
def folders(tmp_path):
    gt_dir, lq_dir = tmp_path / 'gt', tmp_path / 'lq'
    gt_dir.mkdir(), lq_dir.mkdir()
    for i, (h, w) in enumerate([(20, 24), (16, 16), (18, 30)]):
        lq, gt = _coord_pair(h, w, 4)
        Image.fromarray(gt).save(gt_dir / f'im{i}.png')  # stored RGB
        Image.fromarray(lq).save(lq_dir / f'im{i}x4.png')
    return str(gt_dir), str(lq_dir)


def test_path_pairing(folders, tmp_path):
    gt_dir, lq_dir = folders
    paths = paired_paths_from_folder([lq_dir, gt_dir], ['lq', 'gt'], '{}x4')
    assert [os.path.basename(p['gt_path']) for p in paths] == ['im0.png', 'im1.png', 'im2.png']
    assert [os.path.basename(p['lq_path']) for p in paths] == ['im0x4.png', 'im1x4.png', 'im2x4.png']
    with pytest.raises(AssertionError):
        paired_paths_from_folder([lq_dir, gt_dir], ['lq', 'gt'], '{}')  # template does not match the LQ names
    meta = tmp_path / 'meta.txt'
    meta.write_text('im2.png (72,120,3)\nim0.png (80,96,3)\n')
    paths = paired_paths_from_meta_info_file([lq_dir, gt_dir], ['lq', 'gt'], str(meta), '{}x4')
    assert [os.path.basename(p['lq_path']) for p in paths] == ['im2x4.png', 'im0x4.png']


def test_imfrombytes_is_bgr(folders):
    gt_dir, _ = folders
    raw = open(os.path.join(gt_dir, 'im0.png'), 'rb').read()
    rgb = np.asarray(Image.open(os.path.join(gt_dir, 'im0.png')))
    img = imfrombytes(raw)
    assert img.dtype == np.uint8 and np.array_equal(img, rgb[:, :, ::-1])
    f = imfrombytes(raw, float32=True)
    assert f.dtype == np.float32 and np.array_equal(f, img.astype(np.float32) / 255.)
    assert imfrombytes(raw, flag='grayscale').shape == rgb.shape[:2]


def test_paired_random_crop_and_augment_geometry():
    lq, gt = _coord_pair(20, 24, 4)
    random.seed(3)
    st = random.getstate()
    g, l = paired_random_crop(gt, lq, 32, 4)
    random.setstate(st)
    top, left = random.randint(0, 20 - 8), random.randint(0, 24 - 8)  # the documented draw order: top then left
    assert l.shape == (8, 8, 3) and g.shape == (32, 32, 3)
    assert np.array_equal(l, lq[top:top + 8, left:left + 8]) and np.array_equal(g, gt[4 * top:4 * top + 32, 4 * left:4 * left + 32])
    with pytest.raises(ValueError):
        paired_random_crop(gt[:-1], lq, 32, 4)
    with pytest.raises(ValueError):
        paired_random_crop(gt, lq, 128, 4)
    seen = set()
    for seed in range(40):
        random.seed(seed)
        (ga, la), status = augment([g, l], True, True, return_status=True)
        seen.add(status)
        assert np.array_equal(ga, la.repeat(4, 0).repeat(4, 1))  # same symmetry on both images
        ref = l
        if status[0]:
            ref = ref[:, ::-1]
        if status[1]:
            ref = ref[::-1]
        if status[2]:
            ref = ref.transpose(1, 0, 2)
        assert np.array_equal(la, ref) and la.flags['C_CONTIGUOUS']
    assert len(seen) == 8  # all elements of the dihedral group show up
    random.seed(0)
    assert np.array_equal(augment(l, False, False), l)
    assert mod_crop(gt[:79, :93], 4).shape == (76, 92, 3)


def test_paired_random_crop_matches_the_reference(golden):
    """Seeded paired_random_crop / mod_crop against the reference's own transforms.py (golden G-p): same windows for the same
    Python random state, for single images and for lists."""
    g = golden('g_p_crop')
    lq, gt = _coord_pair(20, 24, 4)
    lq, gt = lq.astype(np.float32), gt.astype(np.float32)
    for seed in range(16):
        random.seed(seed)
        gg, ll = paired_random_crop(gt, lq, 32, 4)
        assert np.array_equal(ll, g[f'lq_{seed}']) and np.array_equal(gg[::31, ::31], g[f'gt_corner_{seed}'])
    random.seed(99)
    gs, ls = paired_random_crop([gt, gt + 1], [lq, lq + 1], 16, 4)
    assert np.array_equal(ls[0], g['list_lq0']) and np.array_equal(ls[1], g['list_lq1'])
    assert np.array_equal(gs[1][::15, ::15], g['list_gt1_corner'])
    assert tuple(g['mod_crop_shape']) == mod_crop(gt[:79, :93], 4).shape


def test_paired_image_dataset_train_and_val(folders):
    gt_dir, lq_dir = folders
    base = dict(name='t', type='PairedImageDataset', dataroot_gt=gt_dir, dataroot_lq=lq_dir, filename_tmpl='{}x4',
                io_backend=dict(type='disk'), scale=4)
    assert DATASET_REGISTRY.get('PairedImageDataset') is PairedImageDataset
    val = PairedImageDataset(dict(base, phase='val'))
    assert len(val) == 3
    item = val[0]
    assert item['lq'].shape == (3, 20, 24) and item['gt'].shape == (3, 80, 96) and item['lq'].dtype == torch.float32
    lq, _ = _coord_pair(20, 24, 4)
    assert torch.equal(item['lq'], torch.from_numpy(lq.transpose(2, 0, 1).astype(np.float32) / 255.))  # RGB, CHW, [0, 1]
    assert item['gt_path'].endswith('im0.png') and item['lq_path'].endswith('im0x4.png')
    train = PairedImageDataset(dict(base, phase='train', gt_size=32, use_flip=True, use_rot=True))
    random.seed(5)
    for i in range(3):
        item = train[i]
        assert item['lq'].shape == (3, 8, 8) and item['gt'].shape == (3, 32, 32)
        assert torch.equal(item['gt'], item['lq'].repeat_interleave(4, 1).repeat_interleave(4, 2))
    norm = PairedImageDataset(dict(base, phase='val', mean=[0.5, 0.5, 0.5], std=[0.25, 0.5, 1.0]))[1]
    ref = val[1]
    std = torch.tensor([0.25, 0.5, 1.0]).view(3, 1, 1)
    assert torch.allclose(norm['lq'], (ref['lq'] - 0.5) / std) and torch.allclose(norm['gt'], (ref['gt'] - 0.5) / std)


def test_single_image_dataset(folders, tmp_path):
    """LQ-only items for the test pipeline: folder scan in name order, meta-info mode, optional normalisation."""
    _, lq_dir = folders
    ds = SingleImageDataset(dict(name='s', type='SingleImageDataset', dataroot_lq=lq_dir, io_backend=dict(type='disk')))
    assert DATASET_REGISTRY.get('SingleImageDataset') is SingleImageDataset and len(ds) == 3
    item = ds[2]
    assert set(item) == {'lq', 'lq_path'} and item['lq_path'].endswith('im2x4.png') and item['lq'].shape == (3, 18, 30)
    lq, _ = _coord_pair(18, 30, 4)
    assert torch.equal(item['lq'], torch.from_numpy(lq.transpose(2, 0, 1).astype(np.float32) / 255.))
    meta = tmp_path / 'm.txt'
    meta.write_text('im1x4.png (16,16,3)\n')
    one = SingleImageDataset(dict(name='s', dataroot_lq=lq_dir, meta_info_file=str(meta), io_backend=dict(type='disk'),
                                  mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5]))
    assert len(one) == 1 and torch.allclose(one[0]['lq'], (ds[1]['lq'] - 0.5) / 0.5)


def test_file_client_backends(folders):
    from image_restoration_amd.data.file_client import FileClient
    gt_dir, _ = folders
    p = os.path.join(gt_dir, 'im1.png')
    assert FileClient('disk').get(p) == open(p, 'rb').read()
    with pytest.raises(ValueError):
        FileClient('memcached')
    try:
        import lmdb  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):  # loud, not a silent switch to another backend
            FileClient('lmdb', db_paths=['a.lmdb'], client_keys=['gt'])


def test_prefetchers_iterate_a_loader(folders):
    gt_dir, lq_dir = folders
    ds = PairedImageDataset(dict(name='t', type='PairedImageDataset', dataroot_gt=gt_dir, dataroot_lq=lq_dir, filename_tmpl='{}x4',
                                 io_backend=dict(type='disk'), scale=4, phase='train', gt_size=32, use_flip=False, use_rot=False))
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, drop_last=False)
    for pf in (CPUPrefetcher(loader), CUDAPrefetcher(loader, dict(num_gpu=0))):
        for _ in range(2):  # two epochs through reset()
            pf.reset()
            sizes = []
            while (b := pf.next()) is not None:
                sizes.append(b['lq'].shape[0])
                assert b['gt'].shape[1:] == (3, 32, 32) and len(b['gt_path']) == b['lq'].shape[0]
            assert sizes == [2, 1]


def test_path_pairing_matches_the_reference(golden, tmp_path):
    """paired_paths_from_folder / _from_meta_info_file against the reference's data_util.py on the same file names (golden G-s).
    Folder mode: the reference returns the pairs in os.scandir order (file-system dependent), this build sorts them — the
    pairs themselves must be the same set; meta-info mode keeps the file's order in both."""
    import json
    g = golden('g_s_paths')
    gt_dir, lq_dir = tmp_path / 'gt', tmp_path / 'lq'
    gt_dir.mkdir(), lq_dir.mkdir()
    for n in json.loads(str(g['names'])):
        base, ext = os.path.splitext(n)
        (gt_dir / n).write_bytes(b'')
        (lq_dir / f'{base}x4{ext}').write_bytes(b'')

    def rel(paths):
        return [{k: os.path.relpath(v, str(tmp_path)) for k, v in d.items()} for d in paths]
    mine = rel(paired_paths_from_folder([str(lq_dir), str(gt_dir)], ['lq', 'gt'], '{}x4'))
    ref = json.loads(str(g['folder']))
    key = lambda d: d['gt_path']
    assert sorted(mine, key=key) == sorted(ref, key=key) and mine == sorted(mine, key=key)
    meta = tmp_path / 'meta.txt'
    meta.write_text('a9.png (480,480,3)\n0802.png (480,480,3)\n')
    assert rel(paired_paths_from_meta_info_file([str(lq_dir), str(gt_dir)], ['lq', 'gt'], str(meta), '{}x4')) == json.loads(str(g['meta']))


def test_background_loader_runs_ahead_and_propagates_errors(folders):
    """PrefetchDataLoader / BackgroundIterator (reference prefetch_dataloader.py:7-63): same batches in the same order as the plain
    loader, produced on a daemon thread at most ``num_prefetch_queue`` ahead; a failure inside the producer surfaces in the consumer."""
    from image_restoration_amd.data import BackgroundIterator, PrefetchDataLoader
    gt_dir, lq_dir = folders
    ds = PairedImageDataset(dict(name='t', type='PairedImageDataset', dataroot_gt=gt_dir, dataroot_lq=lq_dir, filename_tmpl='{}x4',
                                 io_backend=dict(type='disk'), scale=4, phase='val'))
    plain = [b['lq_path'] for b in torch.utils.data.DataLoader(ds, batch_size=1)]
    ahead = [b['lq_path'] for b in PrefetchDataLoader(num_prefetch_queue=2, dataset=ds, batch_size=1)]
    assert ahead == plain and len(ahead) == 3
    assert list(BackgroundIterator(iter(range(7)), 3)) == list(range(7))

    def broken():
        yield 1
        raise KeyError('decode failed')
    it = BackgroundIterator(broken(), 2)
    assert next(it) == 1
    with pytest.raises(KeyError):
        next(it)


@pytest.mark.parametrize('mode', [True, 'full'])
def test_device_augment_items_carry_the_same_draws_as_the_host_pipeline(tmp_path, mode):
    """device_augment: the dataset consumes Python's random exactly like the host path (window, then symmetry), so after either
    kind of item the generator is in the same state; the uint8 windows it ships are the host path's crops before any float work."""
    gt_dir, lq_dir = tmp_path / 'gt', tmp_path / 'lq'
    gt_dir.mkdir(), lq_dir.mkdir()
    for i in range(3):
        lq, gt = _coord_pair(20, 24, 4)
        Image.fromarray(gt).save(gt_dir / f'im{i}.png')
        Image.fromarray(lq).save(lq_dir / f'im{i}x4.png')
    base = dict(name='t', type='PairedImageDataset', dataroot_gt=str(gt_dir), dataroot_lq=str(lq_dir), filename_tmpl='{}x4',
                io_backend=dict(type='disk'), scale=4, phase='train', gt_size=32, use_flip=True, use_rot=True)
    host, dev = PairedImageDataset(dict(base)), PairedImageDataset(dict(base, device_augment=mode))
    for i in range(3):
        random.seed(40 + i)
        a = host[i]
        after_host = random.random()
        random.seed(40 + i)
        b = dev[i]
        assert random.random() == after_host
        assert b['lq_u8'].dtype == np.uint8 and b['gt_u8'].dtype == np.uint8 and 0 <= int(b['sym']) < 8
        if mode == 'full':
            assert b['lq_u8'].shape == (20, 24, 3) and b['gt_u8'].shape == (80, 96, 3) and b['window'].shape == (2,)
            t, l = (int(v) for v in b['window'])
            lq_win = b['lq_u8'][t:t + 8, l:l + 8]
        else:
            assert b['lq_u8'].shape == (8, 8, 3) and b['gt_u8'].shape == (32, 32, 3) and 'window' not in b
            lq_win = b['lq_u8']
        from image_restoration_amd.data.transforms import apply_symmetry
        want = apply_symmetry(lq_win, int(b['sym']))[:, :, ::-1].transpose(2, 0, 1).astype(np.float32) / 255.
        assert np.array_equal(a['lq'].numpy(), want)

"""Option files, sampler and tiler planning on CPU."""
import os

import numpy as np
import pytest
import torch

from image_restoration_amd.data import EnlargedSampler, SyntheticPairedDataset
from image_restoration_amd.tiling import plan_tiles
from image_restoration_amd.utils.options import dict2str, parse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sampler_matches_reference_streams(golden):
    g = golden('g_m_sampler')

    class DS:
        def __len__(self):
            return 10

    for rank in (0, 1):
        s = EnlargedSampler(DS(), 2, rank, ratio=3)
        assert len(s) == int(g[f'r{rank}_len'])
        for epoch in (0, 1, 5):
            s.set_epoch(epoch)
            assert list(iter(s)) == g[f'r{rank}_e{epoch}'].tolist()
    # ranks partition the enlarged epoch
    a, b = EnlargedSampler(DS(), 2, 0, 3), EnlargedSampler(DS(), 2, 1, 3)
    assert sorted(list(a) + list(b)) == sorted(i % 10 for i in range(30))


@pytest.mark.parametrize('path,is_train', [
    ('options/train/ESRGAN/train_ESRGAN_x4_synthetic.yml', True),
    ('options/train/ESRGAN/train_RRDBNet_PSNR_x4_synthetic.yml', True),
    ('options/train/ESRGAN/train_RRDBNet_PSNR_x4_folders.yml', True),
    ('training_config/train_rrdbnet_esrgan_x4_mi355x.yml', True),
    ('training_config/train_rrdbnet_esrgan_x4_mi355x_bf16.yml', True),
    ('training_config/train_rrdbnet_esrgan_x4_mi355x_bf16_unet.yml', True),
    ('options/test/ESRGAN/test_ESRGAN_x4_woGT.yml', False),
    ('options/test/ESRGAN/test_ESRGAN_x4.yml', False),
])
def test_option_files_parse_like_the_reference(path, is_train, tmp_path):
    opt = parse(os.path.join(ROOT, path), str(tmp_path), is_train=is_train)
    assert opt['is_train'] is is_train and opt['network_g']['type'] == 'RRDBNet' and opt['scale'] == 4
    if is_train:
        assert opt['path']['models'] == os.path.join(str(tmp_path), 'experiments', opt['name'], 'models')
        assert opt['path']['training_states'].endswith('training_states')
        assert opt['datasets']['train']['phase'] == 'train' and opt['datasets']['train']['scale'] == 4
        assert opt['train']['optim_g']['lr'] in (1e-4, 2e-4) and opt['train']['optim_g']['betas'] == [0.9, 0.99]
        assert isinstance(opt['num_gpu'], int)  # 'auto' resolved
    else:
        assert opt['path']['results_root'] == os.path.join(str(tmp_path), 'results', opt['name'])
        assert opt['datasets']['test_1']['phase'] == 'test' and opt['datasets']['test_1']['scale'] == 4
    assert 'network_g' in dict2str(opt)
    dbg = parse(os.path.join(ROOT, path), str(tmp_path), is_train=is_train, debug=True)
    assert dbg['name'].startswith('debug_')
    if is_train:
        assert dbg['logger']['print_freq'] == 1 and dbg['logger']['save_checkpoint_freq'] == 8


def test_synthetic_dataset_contract():
    ds = SyntheticPairedDataset(dict(num_samples=5, gt_size=64, scale=4, seed=3))
    s = ds[2]
    assert s['lq'].shape == (3, 16, 16) and s['gt'].shape == (3, 64, 64) and s['lq'].dtype == torch.float32
    assert torch.equal(ds[2]['gt'], s['gt']) and not torch.equal(ds[3]['gt'], s['gt'])
    assert 0.0 <= float(s['gt'].min()) and float(s['gt'].max()) < 1.0


def test_tile_plan_covers_frame_exactly_once():
    for (h, w, tile, pad) in [(40, 56, 16, 4), (2160, 3840, 512, 16), (7, 5, 16, 4), (512, 512, 512, 16)]:
        cover = np.zeros((h, w), np.int32)
        for (y0, y1, x0, x1), (py0, py1, px0, px1) in plan_tiles(h, w, tile, pad):
            cover[y0:y1, x0:x1] += 1
            assert 0 <= py0 <= y0 < y1 <= py1 <= h and 0 <= px0 <= x0 < x1 <= px1 <= w
            assert y0 - py0 <= pad and py1 - y1 <= pad
        assert (cover == 1).all()
    assert len(plan_tiles(2160, 3840, 512, 16)) == 40  # BASELINE config 5: 5 x 8 cells


def test_official_esrgan_key_map_roundtrip():
    import image_restoration_amd as ira
    from image_restoration_amd.utils import checkpoint as ck
    net = ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, num_feat=16, num_block=2, num_grow_ch=8))
    sd = net.state_dict()
    official = {ck.basicsr_to_official_key(k): v.clone() + 1 for k, v in sd.items()}
    assert 'RRDB_trunk.1.RDB3.conv5.weight' in official and 'trunk_conv.bias' in official and 'upconv2.weight' in official
    assert 'HRconv.weight' in official and 'conv_first.weight' in official and 'conv_last.bias' in official
    assert ck.is_official_esrgan(official) and not ck.is_official_esrgan(sd)
    assert list(ck.convert_official_esrgan(official).keys()) == list(sd.keys())
    ck.load_generator_weights(net, {'module.' + k: v for k, v in official.items()})
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k] + 1) if False else torch.allclose(v, official[ck.basicsr_to_official_key(k)])
    ck.load_generator_weights(net, {'params': {k: v * 0 for k, v in sd.items()}, 'params_ema': sd}, prefer='params')
    assert float(next(net.parameters()).abs().sum()) == 0.0


def test_psnr_host_matches_definition():
    from image_restoration_amd.metrics import calculate_psnr
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (20, 24, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-3, 4, a.shape), 0, 255).astype(np.uint8)
    mse = np.mean((a[4:-4, 4:-4].astype(np.float64) - b[4:-4, 4:-4].astype(np.float64))**2)
    assert abs(calculate_psnr(a, b, 4) - 20 * np.log10(255 / np.sqrt(mse))) < 1e-12
    assert calculate_psnr(a, a, 0) == float('inf')
    assert abs(calculate_psnr(a.transpose(2, 0, 1), b.transpose(2, 0, 1), 4, input_order='CHW') - calculate_psnr(a, b, 4)) < 1e-12
    assert np.isfinite(calculate_psnr(a, b, 0, test_y_channel=True))
    with pytest.raises(ValueError):
        calculate_psnr(a, b, 0, input_order='XYZ')


def test_ssim_host_matches_direct_2d_definition():
    """calculate_ssim (separable, valid region) against a direct 2-D correlation with the outer-product window
    (scipy), which is what the reference's cv2.filter2D(...)[5:-5, 5:-5] computes (psnr_ssim.py:66-79; cv2 is absent
    here, so the reference itself cannot be run: parity of this metric is pinned to the published definition only)."""
    from scipy.signal import correlate2d
    from image_restoration_amd.metrics import calculate_ssim
    rng = np.random.RandomState(3)
    a = rng.randint(0, 256, (40, 52, 3)).astype(np.uint8)
    b = np.clip(a.astype(np.int32) + rng.randint(-20, 21, a.shape), 0, 255).astype(np.uint8)
    g = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2)); g /= g.sum()
    win = np.outer(g, g)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2
    vals = []
    for ch in range(3):
        x, y = a[4:-4, 4:-4, ch].astype(np.float64), b[4:-4, 4:-4, ch].astype(np.float64)
        f = lambda z: correlate2d(z, win, mode='valid')
        mx, my = f(x), f(y)
        sxx, syy, sxy = f(x * x) - mx * mx, f(y * y) - my * my, f(x * y) - mx * my
        vals.append((((2 * mx * my + c1) * (2 * sxy + c2)) / ((mx * mx + my * my + c1) * (sxx + syy + c2))).mean())
    assert abs(calculate_ssim(a, b, 4) - np.mean(vals)) < 1e-12
    assert abs(calculate_ssim(a, a, 0) - 1.0) < 1e-12
    assert abs(calculate_ssim(a.transpose(2, 0, 1), b.transpose(2, 0, 1), 4, input_order='CHW') - np.mean(vals)) < 1e-12


def _tiler_worker(rank, world, port, q):
    import torch.distributed as dist
    from image_restoration_amd.tiling import tiled_forward
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)

    def net(x):  # a stand-in generator with a 3x3 receptive field: x4 nearest upsampling of a box-filtered image
        y = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode='replicate'), 3, 1)
        return torch.nn.functional.interpolate(y, scale_factor=4, mode='nearest')
    img = torch.rand(1, 3, 37, 50, generator=torch.Generator().manual_seed(4))
    for dt in (torch.float32, torch.uint8):
        out = tiled_forward(net, img, tile=16, pad=2, scale=4, max_batch=2, rank=rank, world_size=world, dst=1, out_dtype=dt)
        q.put((rank, str(dt), None if out is None else out.numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_tiler_gathers_the_frame(world):
    """N>1 tile sharding (SURVEY.md §8 e): cells i % world == rank, one gather to `dst`; the assembled frame equals the
    single-process result in fp32 and in the uint8 output convention; other ranks get None."""
    import torch.multiprocessing as mp
    from image_restoration_amd.tiling import quantise_u8, tiled_forward

    def net(x):
        y = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (1, 1, 1, 1), mode='replicate'), 3, 1)
        return torch.nn.functional.interpolate(y, scale_factor=4, mode='nearest')
    img = torch.rand(1, 3, 37, 50, generator=torch.Generator().manual_seed(4))
    ref = tiled_forward(net, img, tile=16, pad=2, scale=4, max_batch=2)
    assert ref.shape == (1, 3, 148, 200)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_tiler_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2 * world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, dt, out in res:
        if rank != 1:
            assert out is None
        elif dt == 'torch.float32':
            assert np.array_equal(out, ref.numpy())
        else:
            assert out.dtype == np.uint8 and np.array_equal(out, quantise_u8(ref).numpy())


def test_calculate_psnr_matches_the_reference(golden):
    """calculate_psnr against the reference's own psnr_ssim.py (golden G-q): crop_border, HWC / CHW, RGB and y channel."""
    from image_restoration_amd.metrics import calculate_psnr
    g = golden('g_q_psnr')
    a, b = g['img1'], g['img2']
    for cb in (0, 4):
        for y in (False, True):
            assert abs(calculate_psnr(a, b, cb, 'HWC', y) - float(g[f'psnr_cb{cb}_y{int(y)}'])) < 1e-9
            assert abs(calculate_psnr(a.transpose(2, 0, 1), b.transpose(2, 0, 1), cb, 'CHW', y) - float(g[f'psnr_chw_cb{cb}_y{int(y)}'])) < 1e-9
    assert calculate_psnr(a, a, 0) == float(g['psnr_same']) == float('inf')



def test_option_parsing_matches_the_reference(golden):
    """options.parse and dict2str against the reference's own options.py run on this repository's option files (golden G-r):
    identical dictionaries in train / test mode and with --debug, identical printed form."""
    import json
    g = golden('g_r_options')
    files = json.loads(str(g['files']))
    for i, (rel, is_train) in enumerate(files):
        for debug in (False, True):
            opt = parse(os.path.join(ROOT, rel), '/srv/run', is_train=is_train, debug=debug)
            ref = json.loads(str(g[f'f{i}_d{int(debug)}_json']))
            mine = json.loads(json.dumps(opt, sort_keys=True))
            if isinstance(ref.get('num_gpu'), int) and isinstance(mine.get('num_gpu'), int):
                mine['num_gpu'] = ref['num_gpu']  # 'auto' resolves to the local device count
            assert mine == ref, (rel, debug, {k: (mine.get(k), ref.get(k)) for k in set(mine) | set(ref) if mine.get(k) != ref.get(k)})
            if not debug:
                assert dict2str(opt) == str(g[f'f{i}_str']), rel


def test_bench_gpus_n_starts_its_own_ranks_without_touching_the_gpu():
    """``python bench.py --gpus 2`` with no launcher in the environment starts torch.distributed.run as a child process (reference
    entry: scripts/dist_train.sh:15-16) and returns its code.  Here there is no GPU, so both ranks stop at bench.py's "needs a GPU"
    check: what is asserted is that the RANKS said so (the parent never asked for the device) and that the failure is ours."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--no-secondary'],
                       capture_output=True, text=True, timeout=300, cwd=root, env=env)
    assert 'starting 2 ranks' in r.stderr and '--nproc-per-node 2' in r.stderr and '--master-addr 127.0.0.1' in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and 'bench.py needs a GPU' in r.stderr

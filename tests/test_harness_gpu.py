"""Tiled inference against the reference applied per padded cell (golden G-k), the inference script's image round
trip, and the training entry point (a few iterations, checkpoint, --auto_resume) on the GPU."""
import os

import numpy as np
import pytest
import torch
import yaml

import image_restoration_amd as ira
from image_restoration_amd.tiling import tiled_forward
from image_restoration_amd.utils import synth

pytestmark = pytest.mark.gpu


def test_tiled_forward_matches_reference_per_cell(cuda, golden):
    g = golden('g_k_tiled')
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=32, num_block=2, num_grow_ch=16)
    net = ira.build_network(dict(type='RRDBNet', **cfg)).to(cuda).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(91, **cfg).items()}, strict=True)
    img = torch.from_numpy(g['img']).to(cuda)
    out = tiled_forward(net, img, tile=16, pad=4, scale=4, max_batch=3)
    assert float(np.abs(out.cpu().numpy() - g['out']).max()) < 1e-4
    # sharding rule: the union of the ranks' cells is the frame (computed serially here, rank by rank)
    from image_restoration_amd.tiling import _run_cells, plan_tiles
    cells = list(enumerate(plan_tiles(40, 56, 16, 4)))
    crops = {}
    for rank in range(3):
        crops.update(_run_cells(net, img, [c for c in cells if c[0] % 3 == rank], 4, 8))
    assert sorted(crops) == list(range(len(cells)))
    for i, ((y0, y1, x0, x1), _) in cells:
        assert torch.equal(crops[i], out[:, :, y0 * 4:y1 * 4, x0 * 4:x1 * 4])


def test_inference_script_roundtrip(cuda, golden, tmp_path):
    from image_restoration_amd import inference
    g = golden('g_d_c1')
    src = tmp_path / 'crop.png'
    inference.imwrite_bgr(str(src), g['img_u8'])
    assert np.array_equal(inference.imread_bgr(str(src)), g['img_u8'])
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=32, num_block=1, num_grow_ch=32)
    ck = tmp_path / 'net_g.pth'
    torch.save({'params': {'module.' + k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(0, **cfg).items()}}, ck)
    inference.main(['--input', str(src), '--output', str(tmp_path / 'out.png'), '--model_path', str(ck), '--num_feat', '32',
                    '--num_block', '1'])
    out = inference.imread_bgr(str(tmp_path / 'out.png'))
    diff = np.abs(out.astype(np.int32) - g['out_u8'].astype(np.int32))
    assert out.shape == (256, 256, 3) and diff.max() <= 1 and (diff > 0).mean() < 1e-3
    # the reduced-precision kernels behind the same script: a few grey levels at most on this 1-block net
    inference.main(['--input', str(src), '--output', str(tmp_path / 'out16.png'), '--model_path', str(ck), '--num_feat', '32',
                    '--num_block', '1', '--compute_dtype', 'bf16'])
    d16 = np.abs(inference.imread_bgr(str(tmp_path / 'out16.png')).astype(np.int32) - g['out_u8'].astype(np.int32))
    assert d16.max() <= 6 and d16.mean() < 0.5


def test_train_entry_point_runs_saves_and_resumes(cuda, tmp_path):
    from image_restoration_amd.train import train_pipeline
    opt = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'options', 'train',
                                           'ESRGAN', 'train_ESRGAN_x4_synthetic.yml')))
    opt['name'] = 'tiny'
    opt['datasets']['train'].update(num_samples=16, batch_size_per_gpu=2, num_worker_per_gpu=0, dataset_enlarge_ratio=1)
    opt['network_g'].update(num_feat=16, num_block=1, num_grow_ch=8)
    opt['network_d']['num_feat'] = 8
    opt['train']['total_iter'] = 6
    opt['train']['scheduler']['milestones'] = [4]
    opt['logger'].update(print_freq=2, save_checkpoint_freq=4)
    opt['datasets']['val'] = dict(name='synthetic_val', type='SyntheticPairedDataset', num_samples=2, gt_size=64)
    opt['val'] = dict(val_freq=4, save_img=False, metrics=dict(psnr=dict(type='calculate_psnr', crop_border=4, test_y_channel=False),
                                                               ssim=dict(type='calculate_ssim', crop_border=4, test_y_channel=False)))
    p = tmp_path / 'opt.yml'
    yaml.safe_dump(opt, open(p, 'w'))
    model = train_pipeline(str(tmp_path), ['-opt', str(p)])
    exp = tmp_path / 'experiments' / 'tiny'
    assert (exp / 'models' / 'net_g_4.pth').exists() and (exp / 'models' / 'net_d_4.pth').exists()
    assert (exp / 'training_states' / '4.state').exists() and (exp / 'models' / 'net_g_latest.pth').exists()
    log = model.get_current_log()
    # the option file carries the reference's full generator loss: pixel + perceptual (VGG19 conv5_4) + relativistic GAN
    assert set(log) == {'l_g_pix', 'l_g_percep', 'l_g_gan', 'l_d_real', 'l_d_fake', 'out_d_real', 'out_d_fake'}
    assert all(np.isfinite(v) for v in log.values())
    assert abs(model.get_current_learning_rate()[0] - 5e-5) < 1e-12  # milestone 4 halved 1e-4
    assert np.isfinite(model.metric_results['psnr']) and 0 < model.metric_results['ssim'] <= 1  # validation ran (val_freq, end)
    # resume from iteration 4 and finish: same optimiser step count as the uninterrupted run
    model2 = train_pipeline(str(tmp_path), ['-opt', str(p), '--auto_resume'])
    assert model2.optimizer_g.step_count == 6 and model2.optimizer_d.step_count == 6
    ck = torch.load(exp / 'models' / 'net_g_latest.pth', weights_only=False)
    assert set(ck) == {'params', 'params_ema'}


def test_device_psnr_matches_host(cuda):
    from image_restoration_amd.metrics import calculate_psnr, psnr_device
    from image_restoration_amd.utils.img_util import tensor2img
    sr = torch.from_numpy(synth.signed_input(1, (3, 3, 40, 52), 0.7) + 0.5).clamp(-0.2, 1.2)
    gt = torch.from_numpy(synth.uniform_input(2, (3, 3, 40, 52)))
    got = psnr_device(sr.to(cuda), gt.to(cuda), crop_border=4)
    for i in range(3):
        a = tensor2img(sr[i:i + 1], rgb2bgr=True, min_max=(0, 1))
        b = tensor2img(gt[i:i + 1], rgb2bgr=True, min_max=(0, 1))
        assert abs(got[i] - calculate_psnr(a, b, 4)) < 1e-4
    assert psnr_device(gt.to(cuda), gt.to(cuda))[0] == float('inf')


def test_device_ssim_matches_host(cuda):
    from image_restoration_amd.metrics import calculate_ssim, ssim_device
    from image_restoration_amd.utils.img_util import tensor2img
    g = torch.Generator().manual_seed(11)
    gt = torch.rand(2, 3, 48, 64, generator=g)
    sr = (gt + 0.08 * torch.randn(2, 3, 48, 64, generator=g)).clamp(-0.1, 1.1)
    got = ssim_device(sr.to(cuda), gt.to(cuda), crop_border=4)
    for i in range(2):
        a, b = tensor2img(sr[i:i + 1], rgb2bgr=True, min_max=(0, 1)), tensor2img(gt[i:i + 1], rgb2bgr=True, min_max=(0, 1))
        assert abs(got[i] - calculate_ssim(a, b, 4)) < 2e-5, (got[i], calculate_ssim(a, b, 4))
    assert abs(ssim_device(gt.to(cuda), gt.to(cuda))[0] - 1.0) < 1e-6


def test_sr_model_validation_loop_device_metrics(cuda, tmp_path):
    """SRModel.validation (reference sr_model.py:131-184): metric averages over a loader equal the host metrics of the
    tensor2img images of the same outputs; y-channel options take the host route; images are written when asked."""
    from image_restoration_amd.models import build_model
    from image_restoration_amd.metrics import calculate_psnr, calculate_ssim
    from image_restoration_amd.utils.img_util import tensor2img
    opt = dict(name='val_test', model_type='SRModel', scale=4, num_gpu=1, dist=False, rank=0, world_size=1, is_train=True,
               network_g=dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=4, num_feat=16, num_block=1, num_grow_ch=8),
               path=dict(pretrain_network_g=None, strict_load_g=True, visualization=str(tmp_path)),
               train=dict(ema_decay=0, optim_g=dict(type='Adam', lr=1e-4, weight_decay=0, betas=[0.9, 0.99]),
                          scheduler=dict(type='MultiStepLR', milestones=[10], gamma=0.5), total_iter=10, warmup_iter=-1,
                          pixel_opt=dict(type='L1Loss', loss_weight=1.0, reduction='mean')),
               val=dict(metrics=dict(psnr=dict(type='calculate_psnr', crop_border=4, test_y_channel=False),
                                     ssim=dict(type='calculate_ssim', crop_border=4, test_y_channel=False),
                                     psnr_y=dict(type='calculate_psnr', crop_border=4, test_y_channel=True))))
    model = build_model(opt)

    class DS(torch.utils.data.Dataset):
        opt = {'name': 'synthetic_val'}

        def __len__(self):
            return 3

        def __getitem__(self, i):
            g = torch.Generator().manual_seed(i)
            return {'lq': torch.rand(3, 16, 20, generator=g), 'gt': torch.rand(3, 64, 80, generator=g), 'lq_path': f'/x/img{i}.png'}
    loader = torch.utils.data.DataLoader(DS(), batch_size=1, shuffle=False)
    model.validation(loader, 7, None, save_img=True)
    ref = dict(psnr=0.0, ssim=0.0, psnr_y=0.0)
    for data in loader:
        model.feed_data(data)
        model.test()
        a, b = tensor2img([model.output.cpu()]), tensor2img([data['gt']])
        ref['psnr'] += calculate_psnr(a, b, 4) / 3
        ref['ssim'] += calculate_ssim(a, b, 4) / 3
        ref['psnr_y'] += calculate_psnr(a, b, 4, test_y_channel=True) / 3
    assert abs(model.metric_results['psnr'] - ref['psnr']) < 1e-4
    assert abs(model.metric_results['ssim'] - ref['ssim']) < 2e-5
    assert abs(model.metric_results['psnr_y'] - ref['psnr_y']) < 1e-5  # float32 y-channel arithmetic of the reference, summed in a different order
    assert os.path.exists(os.path.join(str(tmp_path), 'img1', 'img1_7.png'))


def test_train_on_png_folders_with_cuda_prefetch(cuda, tmp_path):
    """PairedImageDataset (disk backend) -> EnlargedSampler loader -> CUDAPrefetcher -> SRModel steps; validation runs on
    full images from the same folders.  The prefetcher's device batches equal the loader's host batches."""
    from PIL import Image
    from image_restoration_amd.data import CUDAPrefetcher, PairedImageDataset
    from image_restoration_amd.train import train_pipeline
    rng = np.random.default_rng(0)
    (tmp_path / 'gt').mkdir(), (tmp_path / 'lq').mkdir()
    for i in range(4):
        gt = rng.integers(0, 256, (96, 112, 3), dtype=np.uint8)
        Image.fromarray(gt).save(tmp_path / 'gt' / f'{i:04d}.png')
        Image.fromarray(gt.reshape(24, 4, 28, 4, 3).mean((1, 3)).astype(np.uint8)).save(tmp_path / 'lq' / f'{i:04d}.png')
    ds_opt = dict(name='pngs', type='PairedImageDataset', dataroot_gt=str(tmp_path / 'gt'), dataroot_lq=str(tmp_path / 'lq'),
                  filename_tmpl='{}', io_backend=dict(type='disk'), scale=4, phase='train', gt_size=64, use_flip=True, use_rot=True)
    loader = torch.utils.data.DataLoader(PairedImageDataset(ds_opt), batch_size=2, shuffle=False, pin_memory=True)
    import random
    random.seed(1)
    host = [b for b in loader]
    random.seed(1)
    pf = CUDAPrefetcher(loader, dict(num_gpu=1))
    for hb in host:
        db = pf.next()
        assert db['lq'].is_cuda and torch.equal(db['lq'].cpu(), hb['lq']) and torch.equal(db['gt'].cpu(), hb['gt'])
    assert pf.next() is None
    opt = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'options', 'train',
                                           'ESRGAN', 'train_RRDBNet_PSNR_x4_synthetic.yml')))
    opt['name'] = 'pngs'
    opt['datasets']['train'] = dict(ds_opt, num_worker_per_gpu=2, batch_size_per_gpu=2, dataset_enlarge_ratio=2, prefetch_mode='cuda')
    del opt['datasets']['train']['phase']
    opt['datasets']['val'] = dict(ds_opt, name='pngs_val')
    del opt['datasets']['val']['phase']
    opt['network_g'].update(num_feat=16, num_block=1, num_grow_ch=8)
    opt['train']['total_iter'] = 5
    opt['train']['scheduler'] = dict(type='MultiStepLR', milestones=[4], gamma=0.5)
    opt['logger'].update(print_freq=2, save_checkpoint_freq=100)
    opt['val']['val_freq'] = 100
    p = tmp_path / 'opt.yml'
    yaml.safe_dump(opt, open(p, 'w'))
    model = train_pipeline(str(tmp_path), ['-opt', str(p)])
    assert model.optimizer_g.step_count == 5 and np.isfinite(model.get_current_log()['l_pix'])
    assert np.isfinite(model.metric_results['psnr']) and 0 < model.metric_results['ssim'] <= 1


def test_test_entry_point_saves_images_and_metrics(cuda, tmp_path):
    """python -m image_restoration_amd.test: a paired dataset (metrics + images) and an LQ-only dataset (images) through
    SRModel.validation in test mode; the saved PNG equals the network output in the tensor2img convention."""
    from PIL import Image
    from image_restoration_amd.test import test_pipeline
    from image_restoration_amd.utils.img_util import tensor2img
    rng = np.random.default_rng(1)
    (tmp_path / 'gt').mkdir(), (tmp_path / 'lq').mkdir()
    for i in range(2):
        gt = rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)
        Image.fromarray(gt).save(tmp_path / 'gt' / f'p{i}.png')
        Image.fromarray(gt.reshape(16, 4, 20, 4, 3).mean((1, 3)).astype(np.uint8)).save(tmp_path / 'lq' / f'p{i}.png')
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=16, num_block=1, num_grow_ch=8)
    ck = tmp_path / 'net_g.pth'
    torch.save({'params': {k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(5, **cfg).items()}}, ck)
    opt = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'options', 'test', 'ESRGAN',
                                           'test_ESRGAN_x4.yml')))
    opt['name'] = 'tiny_test'
    opt['datasets'] = dict(test_1=dict(name='pairs', type='PairedImageDataset', dataroot_gt=str(tmp_path / 'gt'),
                                       dataroot_lq=str(tmp_path / 'lq'), io_backend=dict(type='disk')),
                           test_2=dict(name='lq_only', type='SingleImageDataset', dataroot_lq=str(tmp_path / 'lq'),
                                       io_backend=dict(type='disk')))
    opt['network_g'].update(num_feat=16, num_block=1, num_grow_ch=8)
    opt['path'].update(pretrain_network_g=str(ck))
    opt['val']['suffix'] = 'x4'
    p = tmp_path / 'test.yml'
    yaml.safe_dump(opt, open(p, 'w'))
    model = test_pipeline(str(tmp_path), ['-opt', str(p)])
    vis = tmp_path / 'results' / 'tiny_test' / 'visualization'
    for ds in ('pairs', 'lq_only'):
        assert sorted(os.listdir(vis / ds)) == ['p0_x4.png', 'p1_x4.png']
    net = ira.build_network(dict(type='RRDBNet', **cfg)).to(cuda).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(5, **cfg).items()})
    lq = torch.from_numpy(np.asarray(Image.open(tmp_path / 'lq' / 'p1.png')).transpose(2, 0, 1).astype(np.float32) / 255.)[None]
    with torch.no_grad():
        want = tensor2img([net(lq.to(cuda)).cpu()], rgb2bgr=False)
    assert np.array_equal(np.asarray(Image.open(vis / 'lq_only' / 'p1_x4.png')), want)
    assert not hasattr(model, 'optimizer_g')  # test mode builds no training state


def test_inference_script_sharded_over_two_processes(cuda, tmp_path):
    """python -m torch.distributed.run ... inference --launcher pytorch --tile: two ranks (gloo rendezvous, sharing this GPU) split
    the tiles of a frame; rank 0's assembled PNG is bit-identical to the single-process tiled result."""
    import subprocess
    import sys
    from PIL import Image
    from image_restoration_amd import inference
    rng = np.random.default_rng(3)
    Image.fromarray(rng.integers(0, 256, (70, 90, 3), dtype=np.uint8)).save(tmp_path / 'a.png')
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=32, num_block=1, num_grow_ch=32)
    torch.save({'params': {k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(1, **cfg).items()}}, tmp_path / 'g.pth')
    common = ['--input', str(tmp_path / 'a.png'), '--model_path', str(tmp_path / 'g.pth'), '--num_feat', '32', '--num_block', '1',
              '--tile', '32', '--tile_pad', '8']
    inference.main(common + ['--output', str(tmp_path / 'single.png')])
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(29600 + os.getpid() % 300), '-m', 'image_restoration_amd.inference', '--launcher', 'pytorch',
                        '--dist_backend', 'gloo', '--output', str(tmp_path / 'dist.png')] + common,
                       capture_output=True, text=True, timeout=300, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    assert np.array_equal(np.asarray(Image.open(tmp_path / 'single.png')), np.asarray(Image.open(tmp_path / 'dist.png')))


def test_bench_line_contract(cuda):
    """bench.py prints exactly one JSON line with the driver's fields, the roofline object of the dominant kernel, the CPU baseline and
    the secondary workloads (BASELINE configs 3 and 5, each with its own roofline) — checked on a short run."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--steps', '2', '--warmup', '1'], capture_output=True, text=True,
                       timeout=900, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
              'data', 'config', 'roofline', 'cpu_baseline', 'secondary'):
        assert k in d, k
    assert d['unit'] == 'images/sec' and d['n_gpus'] == 1 and d['steps'] == 2 and d['warmup'] == 1 and d['higher_is_better'] is True
    assert d['scaling'] == 'weak' and d['vs_baseline'] is None and d['dtype'] == 'f32' and d['data'] == 'synthetic'
    assert 'workload' in d['config'] and 'model' not in d['config']
    assert abs(d['value'] - 16 * 2 / (d['ms_per_step'] * 2 / 1e3)) < 0.05 * d['value']

    def check_roofline(rf):
        assert rf['bound'] in ('hbm', 'mfma') and rf['unit'] in ('GB/s', 'TFLOP/s') and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3
        assert 0.05 < rf['frac'] < 1.0 and (rf['traffic'] is None or rf['traffic'] > 1e6) and 'stored' in rf['traffic_source']
    check_roofline(d['roofline'])
    assert d['roofline']['frac'] > 0.5
    cb = d['cpu_baseline']
    assert cb['kind'] in ('port', 'reference') and cb['cores'] >= 1 and cb['value'] >= cb['median'] > 0 and cb['unit'] == 'images/sec'
    assert cb['sample'] and sum(len(v) for v in cb['all_rates'].values()) >= 3
    sec = d['secondary']
    assert set(sec) == {'c2_infer_bf16', 'c3_train_step', 'c5_tiled_4k_bf16', 'c5_tiled_4k_fp32', 'recipe_train_step_bf16',
                        'recipe_train_step_fp32'} and not any('error' in v for v in sec.values()), sec
    c2 = sec['c2_infer_bf16']
    assert c2['unit'] == 'images/sec' and c2['dtype'] == 'bf16' and c2['value'] > 500 and 'rdb_fused' in c2['roofline']['kernel']
    check_roofline(c2['roofline'])
    c3 = sec['c3_train_step']
    assert c3['unit'] == 'images/sec' and c3['config']['global_batch'] == 32 and 'UNetDiscriminatorSN' in c3['metric'] and c3['value'] > 50
    assert all(np.isfinite(v) for v in c3['losses'].values())
    check_roofline(c3['roofline'])
    for key in ('c5_tiled_4k_bf16', 'c5_tiled_4k_fp32'):
        assert sec[key]['unit'] == 'frames/sec' and sec[key]['scaling'] == 'strong' and sec[key]['value'] > 0.1
    check_roofline(sec['c5_tiled_4k_bf16']['roofline'])


def test_bench_gpus_2_as_one_command(cuda):
    """``python bench.py --gpus 2 ...`` typed as ONE command (no external launcher): the parent starts torch.distributed.run itself
    before touching the GPU (reference entry: scripts/dist_train.sh:15-16, utils/dist_util.py:21-25).  Two gloo ranks share this
    box's one GPU (SR_BENCH_BACKEND=gloo; on a multi-GPU node the default backend is RCCL): rc 0, one JSON line, n_gpus == 2, the
    process group's own view of the ranks, all six secondaries, no dense-block launch fell back.  The line is kept under
    gpurun_out/ for profiles/."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SR_BENCH_BACKEND='gloo')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--secondary-timeout', '900'], capture_output=True, text=True, timeout=1100, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert 'starting 2 ranks' in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    try:
        os.makedirs(os.path.join(root, 'gpurun_out'), exist_ok=True)
        with open(os.path.join(root, 'gpurun_out', 'bench_gpus2_one_command.json'), 'w') as f:
            f.write(lines[0] + '\n')
    except OSError:
        pass
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['scaling'] == 'weak' and d['config']['global_batch'] == 32
    assert abs(d['value'] - 32 * 2 / (d['ms_per_step'] * 2 / 1e3)) < 0.05 * d['value']
    rv = d['ranks']
    assert rv['world_size'] == 2 and rv['backend'] == 'gloo' and [p['rank'] for p in rv['per_rank']] == [0, 1]
    assert rv['images_per_sec_min'] > 0 and rv['images_per_sec_max'] >= rv['images_per_sec_min']
    sec = d['secondary']
    assert set(sec) == {'c2_infer_bf16', 'c3_train_step', 'c5_tiled_4k_bf16', 'c5_tiled_4k_fp32', 'recipe_train_step_bf16',
                        'recipe_train_step_fp32'} and not any('error' in v for v in sec.values()), sec
    assert sec['c3_train_step']['config']['global_batch'] == 64 and sec['c3_train_step']['n_gpus'] == 2
    for key in ('c5_tiled_4k_bf16', 'c5_tiled_4k_fp32'):
        assert sec[key]['handoff_fallbacks'] == 0 and sec[key]['n_gpus'] == 2


@pytest.mark.parametrize('dtype', ['bf16', 'fp32'])
def test_tiled_forward_at_the_4k_cell_size_equals_the_untiled_forward_of_every_padded_cell(cuda, dtype):
    """BASELINE config 5's geometry (512 x 512 LR cells + 16 px pad, 23 blocks, nf 64) on a 1024 x 1024 frame: each assembled
    2048 x 2048 HR crop equals, bit for bit, the centre of the un-tiled forward of its own padded cell run alone (the
    paste rule; G-k pins the rule at tile 16 against the reference, this pins the product at the size it is benchmarked at — the
    batched cells go through the fused dense-block kernel as a sliding window of 544 tiles per image in bf16)."""
    from image_restoration_amd import watchdog
    from image_restoration_amd.tiling import plan_tiles
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32)
    net = ira.build_network(dict(type='RRDBNet', compute_dtype=dtype, **cfg)).to(cuda).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(0, **cfg).items()}, strict=True)
    img = torch.from_numpy(synth.uniform_input(21, (1, 3, 1024, 1024))).to(cuda)
    before = watchdog.fallback_count
    out = tiled_forward(net, img, tile=512, pad=16, scale=4)
    assert out.shape == (1, 3, 4096, 4096) and watchdog.fallback_count == before
    cells = plan_tiles(1024, 1024, 512, 16)
    assert len(cells) == 4
    with torch.no_grad():
        for (y0, y1, x0, x1), (py0, py1, px0, px1) in cells:
            assert (py1 - py0, px1 - px0) == (528, 528)
            alone = net(img[:, :, py0:py1, px0:px1].contiguous())
            oy, ox = (y0 - py0) * 4, (x0 - px0) * 4
            assert torch.equal(out[:, :, y0 * 4:y1 * 4, x0 * 4:x1 * 4], alone[:, :, oy:oy + 2048, ox:ox + 2048]), (y0, x0)
    assert not watchdog.tripped()
    assert torch.isfinite(out).all() and float(out.std()) > 0

"""Data-parallel training on the HIP path: replicas start identical, stay identical, and two ranks at batch B equal one rank at 2B.

The reference gets this from DistributedDataParallel (basicsr/models/base_model.py:62-76): its constructor broadcasts rank 0's
parameters and buffers, its backward averages gradients.  Here it is BaseModel.align_replicas + NetPack.update (one arena all-reduce),
tested with two processes sharing the GPU (gloo) that seed themselves ``manual_seed + rank`` like utils/options.py — i.e. from an
UNSEEDED-equal start, every randomly initialised tensor differs between ranks before alignment."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))


def _launch(tmp_path, model, iters=3):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(29700 + os.getpid() % 200), os.path.join(ROOT, 'tests', 'helpers', 'dp_worker.py'),
                        '--model', model, '--out', str(tmp_path), '--iters', str(iters)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return [dict(np.load(os.path.join(tmp_path, f'rank{k}.npz'))) for k in range(2)]


@pytest.mark.parametrize('model', ['SRModel', 'ESRGANModel', 'ESRGANModel_unet'])
def test_two_ranks_hold_bit_identical_replicas_after_every_step(cuda, tmp_path, model):
    """Ranks seeded differently (their RNG probes differ) build G, D (BatchNorm VGG or spectral-norm UNet) and the EMA shadow from
    scratch; after construction and after each of 3 optimize_parameters every parameter arena, the EMA arena and every buffer —
    BatchNorm running statistics and counters, spectral-norm u / v — are bit-identical on the two ranks at the start of a step, and
    parameters (which only see all-reduced gradients) after every step."""
    r0, r1 = _launch(tmp_path, model)
    assert not np.array_equal(r0['probe'], r1['probe'])
    keys = [k for k in r0 if k.startswith('it')]
    assert any('_d_params' in k for k in keys) == (model != 'SRModel')
    for k in keys:
        step, what = k.split('_', 1)
        per_rank_stat = '_buf_' in what and ('running_' in what) and step in ('it1', 'it2', 'it3')
        if per_rank_stat:
            continue  # BatchNorm running statistics follow each rank's own batches between two refreshes (DDP: same)
        assert np.array_equal(r0[k], r1[k]), k
    if model == 'ESRGANModel_unet':
        assert any('weight_u' in k for k in keys)   # spectral-norm vectors were among the compared buffers
    assert np.all(np.isfinite(r0['logs']))


def test_two_ranks_at_batch_b_equal_one_rank_at_batch_2b(cuda, tmp_path):
    """BatchNorm-free SRModel (L1): gradient of the global mean == mean of the rank gradients.  The summed gradient arena after the
    first exchange is compared to a single-process backward over the concatenated batch (fp32 summation order only: 1e-5 of the
    largest entry), and after 3 Adam steps the parameters agree to 1e-3 relative L2 (Adam turns rounding-level gradient
    differences of near-zero entries into +-lr, so the bound is on the norm, not per element)."""
    from dp_worker import global_batch, options
    from image_restoration_amd.models import build_model
    r0, _ = _launch(tmp_path, 'SRModel')
    model = build_model(options('SRModel', 0, 1, False))
    model.net_g.load_state_dict(torch.load(os.path.join(tmp_path, 'g_init.pth')), strict=True)
    model.net_g.invalidate_packed()
    model.model_ema(0)
    assert np.array_equal(model.optimizer_g.flat_p.cpu().numpy(), r0['it0_g_params'])
    for it in range(1, 4):
        model.update_learning_rate(it, warmup_iter=-1)
        lq, gt = global_batch(it, 4)
        model.feed_data({'lq': lq, 'gt': gt})
        model.optimize_parameters(it)
        if it == 1:
            single = model.optimizer_g.flat_g.cpu().numpy()
            summed = r0['it1_g_grad'] * 0.5           # the arena holds the SUM over 2 ranks; DDP's mean is folded into Adam
            assert np.abs(single - summed).max() <= 1e-5 * np.abs(single).max()
        mine, theirs = model.optimizer_g.flat_p.cpu().numpy().astype(np.float64), r0[f'it{it}_g_params'].astype(np.float64)
        assert np.linalg.norm(mine - theirs) <= 1e-3 * np.linalg.norm(theirs), it
    assert abs(model.get_current_log()['l_pix'] - float(r0['logs'][-1][list(r0['log_keys']).index('l_pix')])) < 1e-3


def test_collectives_of_the_path_run_through_rccl_on_device_tensors(cuda, tmp_path):
    """backend 'nccl' = RCCL: one rank (a one-GPU box cannot host two RCCL ranks) runs data-parallel ESRGAN steps — arena all-reduce,
    loss reduce, replica broadcast — and the tiler's gather on DEVICE tensors through the RCCL communicator; parameters after two
    steps equal a single-process run bit for bit (the mean over one rank is the identity)."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''), HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', str(29900 + os.getpid() % 90), os.path.join(ROOT, 'tests', 'helpers', 'rccl_one_rank_worker.py'),
                        '--out', str(tmp_path)], capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    out = json.load(open(tmp_path / 'rccl_one_rank.json'))
    assert out['backend'] == 'nccl' and out['params_equal_single_process'] and out['gather_equal']
    assert all(v == v for v in out['log'].values())

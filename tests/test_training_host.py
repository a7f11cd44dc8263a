"""Host-side training logic on CPU: LR schedules vs the reference's sequences (golden G-j), the flat-arena optimiser's
aliasing / checkpoint format, model construction from a reference-schema option dict, and the N>1 gradient
exchange over gloo (world_size 2)."""
import copy
import os

import numpy as np
import pytest
import torch

import image_restoration_amd as ira
from image_restoration_amd import optim
from image_restoration_amd.models import lr_scheduler


class _Opt:
    def __init__(self, lr):
        self.param_groups = [dict(lr=lr)]


def _run(cls, n, **kw):
    o = _Opt(2e-4)
    s = cls(o, **kw)
    out = []
    for it in range(1, n + 1):
        if it > 1:
            s.step()
        out.append(o.param_groups[0]['lr'])
    return np.array(out)


def test_lr_schedules_match_reference(golden):
    g = golden('g_j_lr')
    np.testing.assert_allclose(_run(lr_scheduler.MultiStepRestartLR, 60, milestones=[10, 20, 40, 50], gamma=0.5),
                               g['multistep'], rtol=1e-12)
    np.testing.assert_allclose(_run(lr_scheduler.MultiStepRestartLR, 60, milestones=[10, 20, 35, 50], gamma=0.5,
                                    restarts=[0, 30], restart_weights=[1, 0.5]), g['multistep_restart'], rtol=1e-12)
    np.testing.assert_allclose(_run(lr_scheduler.CosineAnnealingRestartLR, 40, periods=[10, 10, 10, 10],
                                    restart_weights=[1, 0.5, 0.5, 0.5], eta_min=1e-7), g['cosine'], rtol=1e-12)


def test_flat_adam_arena_aliasing_and_state_dict_format():
    net = ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, num_feat=16, num_block=1, num_grow_ch=8))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    opt = optim.FlatAdam(net.parameters(), 1e-3, betas=(0.9, 0.99), modules=[net])
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    assert net._grad_sink is not None and len(net._grad_sink.grad_ptrs) == len(list(net.parameters()))
    p0 = next(net.parameters())
    assert p0.data_ptr() == opt.flat_p.data_ptr() and p0.grad.data_ptr() == opt.flat_g.data_ptr()
    opt.flat_g.fill_(2.0)
    assert float(p0.grad.sum()) == 2.0 * p0.numel()
    opt.zero_grad()
    assert float(opt.flat_g.abs().sum()) == 0.0 and p0.grad.data_ptr() == opt.flat_g.data_ptr()
    with pytest.raises(ira._lib.SrHipError):
        opt.step()  # no CPU fallback
    # torch.optim.Adam checkpoint format round trip
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in net.parameters()], 1e-3, betas=(0.9, 0.99))
    for p in ref.param_groups[0]['params']:
        p.grad = torch.ones_like(p)
    ref.step()
    sd = ref.state_dict()
    opt.load_state_dict(copy.deepcopy(sd))
    assert opt.step_count == 1
    out = opt.state_dict()
    assert set(out['state'].keys()) == set(sd['state'].keys())
    for i in sd['state']:
        assert torch.allclose(out['state'][i]['exp_avg'], sd['state'][i]['exp_avg'])
        assert torch.allclose(out['state'][i]['exp_avg_sq'], sd['state'][i]['exp_avg_sq'])
    assert out['param_groups'][0]['lr'] == 1e-3 and out['param_groups'][0]['params'] == list(range(len(list(net.parameters()))))


def _opt(model_type):
    from collections import OrderedDict as OD
    opt = OD(name='t', model_type=model_type, scale=4, num_gpu=0, manual_seed=0, is_train=True, dist=False, rank=0, world_size=1)
    opt['network_g'] = OD(type='RRDBNet', num_in_ch=3, num_out_ch=3, num_feat=16, num_block=1, num_grow_ch=8)
    opt['network_d'] = OD(type='VGGStyleDiscriminator128', num_in_ch=3, num_feat=8)
    opt['path'] = OD(pretrain_network_g=None, strict_load_g=True, resume_state=None)
    tr = OD(ema_decay=0.9)
    tr['optim_g'] = OD(type='Adam', lr=1e-3, weight_decay=0, betas=[0.9, 0.99])
    tr['optim_d'] = OD(type='Adam', lr=1e-3, weight_decay=0, betas=[0.9, 0.99])
    tr['scheduler'] = OD(type='MultiStepLR', milestones=[2, 3], gamma=0.5)
    tr['pixel_opt'] = OD(type='L1Loss', loss_weight=1e-2, reduction='mean')
    tr['gan_opt'] = OD(type='GANLoss', gan_type='vanilla', real_label_val=1.0, fake_label_val=0.0, loss_weight=5e-3)
    tr['net_d_iters'] = 1
    tr['net_d_init_iters'] = 0
    opt['train'] = tr
    return opt


def test_models_build_from_reference_schema_and_checkpoint_roundtrip(tmp_path):
    from image_restoration_amd.models import build_model
    opt = _opt('ESRGANModel')
    opt['path']['models'] = str(tmp_path)
    opt['path']['training_states'] = str(tmp_path)
    model = build_model(opt)
    assert type(model).__name__ == 'ESRGANModel' and len(model.optimizers) == 2 and len(model.schedulers) == 2
    assert model.get_current_learning_rate() == [1e-3]
    model.update_learning_rate(2)
    model.update_learning_rate(3)
    assert abs(model.get_current_learning_rate()[0] - 5e-4) < 1e-12
    model.save(0, 5)
    ck = torch.load(os.path.join(tmp_path, 'net_g_5.pth'), weights_only=False)
    assert set(ck.keys()) == {'params', 'params_ema'} and len(ck['params']) == len(model.net_g.state_dict())
    assert 'bn0_1.running_mean' in torch.load(os.path.join(tmp_path, 'net_d_5.pth'), weights_only=False)['params']
    st = torch.load(os.path.join(tmp_path, '5.state'), weights_only=False)
    assert st['iter'] == 5 and len(st['optimizers']) == 2 and len(st['schedulers']) == 2
    model2 = build_model(_opt('ESRGANModel'))
    model2.load_network(model2.net_g, os.path.join(tmp_path, 'net_g_5.pth'), True, 'params_ema')
    model2.resume_training(st)
    assert abs(model2.get_current_learning_rate()[0] - 5e-4) < 1e-12
    with pytest.raises(NotImplementedError):
        bad = _opt('SRModel'); bad['train']['optim_g']['type'] = 'SGD'; build_model(bad)
    with pytest.raises(ValueError):
        bad = _opt('SRModel'); bad['train'].pop('pixel_opt'); build_model(bad)
    with pytest.raises(NotImplementedError):
        bad = _opt('SRModel'); bad['train']['scheduler']['type'] = 'Nope'; build_model(bad)


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    from image_restoration_amd.models import build_model
    from image_restoration_amd.utils.options import set_random_seed
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    set_random_seed(11 + rank)  # utils/options.py: manual_seed + rank, so the ranks draw DIFFERENT initial weights
    probe = float(torch.rand(1))
    opt = _opt('ESRGANModel')
    opt.update(dist=True, rank=rank, world_size=world)
    model = build_model(opt)    # ends with BaseModel.align_replicas(): rank 0's parameters, buffers and EMA shadow everywhere
    digest = []
    for label, pack in model.packs.items():
        for t in pack.state_tensors():
            digest.append((label, tuple(t.shape), float(t.double().sum()), float(t.double().abs().sum())))
    # a buffer that drifted on one rank comes back with refresh_buffers() (DDP broadcast_buffers semantics)
    model.net_d.bn0_1.running_mean.add_(float(rank))
    model.refresh_buffers()
    drift = float(model.net_d.bn0_1.running_mean.abs().max())
    opt_g = model.optimizer_g
    opt_g.zero_grad()
    opt_g.flat_g.fill_(float(rank + 1))  # stands for this rank's local gradient
    scale = opt_g.all_reduce_grads()
    log = model.reduce_loss_dict({'l': torch.tensor(float(rank + 1))})  # reduce to rank 0 then / world (base_model.py:336-347)
    q.put((rank, float(opt_g.flat_g[0]), scale, float(opt_g.flat_g.min()), float(opt_g.flat_g.max()), log['l'], probe, digest, drift))
    dist.destroy_process_group()


def test_data_parallel_replica_alignment_and_gradient_exchange_world2():
    """Two gloo ranks seeded differently build ESRGANModel (G + EMA shadow + BatchNorm discriminator) from scratch: every parameter
    arena, shadow arena and buffer is identical across ranks after construction (what DDP's constructor broadcast gives the
    reference, base_model.py:70-73); the gradient arena is SUM-reduced with the mean folded into the update; logged losses are
    averaged onto rank 0."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, g0, scale, gmin, gmax, l, probe, digest, drift in res:
        assert g0 == 3.0 and gmin == 3.0 and gmax == 3.0 and scale == 0.5  # sum over ranks, mean applied in the update
        assert drift == 0.0                                                # rank 1's +1 drift was overwritten by rank 0's zeros
    assert res[0][6] != res[1][6]                 # the ranks' random streams differ ...
    assert res[0][7] == res[1][7] and len(res[0][7]) > 20   # ... their networks do not
    assert res[0][5] == 1.5  # rank 0 holds the mean of the logged loss

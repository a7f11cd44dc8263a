"""The hand-off watchdog of the fused bf16 dense block where results leave the device (image_restoration_amd/watchdog.py).

The contract being protected is the reference's: a forward returns the right image or raises (basicsr/models/sr_model.py:120-129).
The fused kernel's bounded waits can spin out when the GPU withholds CUs from the process; the kernel then raises an abort word and
what it wrote is invalid.  Covered here: (1) a raised word at any point of a tiled / whole-image inference is noticed before the image
is handed over, the work is repeated on the chain launch in the same process and the result is bit-identical; (2) the kernel's own
give-up path, made deterministic with the development grid cap (fewer workgroups than the window of tile rows the hand-offs need);
(3) two processes sharing the GPU through the fused kernel at the tiler's cell size — the case that timed out with a static tile
assignment (round 2) — bit-identical to a single process, watchdog silent; (4) training raises instead of logging / saving."""
import ctypes as C
import json
import logging
import os
import subprocess
import sys

import pytest
import torch

import image_restoration_amd as ira
from image_restoration_amd import _lib, tiling, watchdog
from image_restoration_amd.utils import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def fused_default():
    lib = _lib.load()
    lib.sr_dev_fused_grid_cap.argtypes = [C.c_int]
    lib.sr_dev_fused_grid_cap.restype = None
    lib.sr_dev_chain_watch.argtypes = [C.c_void_p, C.c_void_p]
    lib.sr_dev_chain_watch.restype = None
    _lib.check(lib.sr_set_conv_chain(3), 'sr_set_conv_chain')
    yield
    lib.sr_dev_fused_grid_cap(0)
    torch.cuda.synchronize()
    lib.sr_chain_watchdog()
    _lib.check(lib.sr_set_conv_chain(3), 'sr_set_conv_chain')


def _net(dev, num_block, dtype='bf16'):
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=num_block, num_grow_ch=32)
    net = ira.build_network(dict(type='RRDBNet', compute_dtype=dtype, **cfg)).to(dev).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(3, **cfg).items()}, strict=True)
    return net


def _raise_word(dev):
    word = torch.ones(4, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    _lib.load().sr_dev_chain_watch(word.data_ptr(), None)   # what a driver queues behind a launch whose waits spun out
    torch.cuda.synchronize()


@pytest.mark.parametrize('when', ['before the first cell', 'behind the last cell'])
def test_tiled_forward_repeats_the_cells_on_the_chain_launch_after_a_timeout(cuda, caplog, monkeypatch, when):
    net = _net(cuda, 2)
    img = torch.from_numpy(synth.uniform_input(5, (1, 3, 96, 160))).to(cuda)
    want = tiling.tiled_forward(net, img, tile=64, pad=16, scale=4, max_batch=2)
    before = watchdog.fallback_count
    if when == 'before the first cell':
        _raise_word(cuda)                 # the first forward of the run is refused by the driver (chain_check on entry)
    else:
        real, calls = tiling._run_cells, []

        def run_then_time_out(*a, **k):   # the LAST launch of the run times out: only the watchdog can see it
            out = real(*a, **k)
            if not calls:
                _raise_word(cuda)
            calls.append(1)
            return out
        monkeypatch.setattr(tiling, '_run_cells', run_then_time_out)
    with caplog.at_level(logging.WARNING, logger='basicsr'):
        got = tiling.tiled_forward(net, img, tile=64, pad=16, scale=4, max_batch=2)
    assert torch.equal(got, want)
    assert watchdog.fallback_count == before + 1
    assert any('timed out' in r.getMessage() and 'chain launch' in r.getMessage() for r in caplog.records)
    assert _lib.load().sr_chain_watchdog() == 0


def test_grid_capped_fused_launch_gives_up_and_the_product_recovers(cuda, caplog):
    """3 workgroups for an image of 4 x 8 tiles: tile 0 waits for tile 4, which nobody can claim — every bounded wait spins out (a
    few seconds), the abort word goes up, the workgroups leave through the give-up path and the launch ends.  The raw forward's
    result is then invalid and the watchdog says so; through watchdog.guarded the same forward returns the right image."""
    lib = _lib.load()
    net = _net(cuda, 1)
    x = torch.from_numpy(synth.uniform_input(9, (1, 3, 128, 128))).to(cuda)
    with torch.no_grad():
        want = net(x)
        torch.cuda.synchronize()
        assert lib.sr_chain_watchdog() == 0
        lib.sr_dev_fused_grid_cap(3)
        net(x)
        assert watchdog.tripped()                      # the kernel gave up by itself and the driver's copy of the word shows it
        assert lib.sr_chain_watchdog() == 0            # reported once
        before = watchdog.fallback_count
        with caplog.at_level(logging.WARNING, logger='basicsr'):
            got = watchdog.guarded(lambda: net(x), 'test')   # times out again (the cap is still on), then the chain launch
        assert watchdog.fallback_count == before + 1 and torch.equal(got, want)
        lib.sr_dev_fused_grid_cap(0)
        _lib.check(lib.sr_set_conv_chain(3), 'sr_set_conv_chain')
        assert torch.equal(net(x), want) and not watchdog.tripped()   # and the fused kernel is sound again afterwards


def _worker(tmp_path, world, reps=4, extra=()):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''), HSA_ENABLE_IPC_MODE_LEGACY='0')
    script = os.path.join(ROOT, 'tests', 'helpers', 'shared_gpu_worker.py')
    tail = [script, '--out', str(tmp_path), '--world', str(world), '--reps', str(reps), *extra]
    cmd = [sys.executable] + tail if world == 1 else \
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
         '--master-port', str(29400 + os.getpid() % 200)] + tail
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    return [json.load(open(os.path.join(tmp_path, f'shared_rank{k}_of{world}.json'))) for k in range(world)]


def test_two_processes_sharing_the_gpu_run_the_fused_dense_block_on_tiler_cells(cuda, tmp_path):
    """Fresh children (no GPU call before torch.distributed.run starts them).  Each runs the 23-block bf16 generator on 2 x 544 x 544
    cells four times, concurrently with the other; every output equals the single-process output bit for bit and no launch timed out."""
    single = _worker(tmp_path, 1, reps=1)[0]
    assert not single['tripped'] and not single['errors'] and len(single['digests']) == 1
    ranks = _worker(tmp_path, 2)
    for r in ranks:
        assert not r['errors'] and not r['tripped'] and r['fallbacks'] == 0, r
        assert r['digests'] == [single['digests'][0]] * 4, r['rank']
    (tmp_path / 'summary.json').write_text(json.dumps(dict(single=single, ranks=ranks)))
    keep = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(keep):                                      # a kept record of the run
        json.dump(dict(single=single, ranks=ranks), open(os.path.join(keep, 'shared_gpu_fused.json'), 'w'))


def test_training_step_raises_when_a_launch_timed_out(cuda):
    """A time-out of a step's launches raises at the next driver entry / hand-over (optimize_parameters, get_current_log, save)
    instead of being logged as a step, and save() refuses to write a checkpoint behind it."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
    from dp_worker import global_batch, options
    from image_restoration_amd.models import build_model
    model = build_model(options('SRModel', 0, 1, False, bf16=True))
    lq, gt = global_batch(1, 2)
    model.feed_data({'lq': lq, 'gt': gt})
    model.optimize_parameters(1)
    _raise_word(cuda)
    model.feed_data({'lq': lq, 'gt': gt})
    with pytest.raises(_lib.SrHipError, match='timed out'):
        model.optimize_parameters(2)
    _raise_word(cuda)
    with pytest.raises(_lib.SrHipError, match='timed out'):
        model.save(0, 2)


def test_gan_checkpoint_writes_no_discriminator_file_behind_a_timed_out_step(cuda, tmp_path):
    """SRGANModel.save (ESRGANModel's too) checks the watchdog before its FIRST file: no net_d_<iter>.pth of an invalid step."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
    from dp_worker import global_batch, options
    from image_restoration_amd.models import build_model
    opt = options('ESRGANModel', 0, 1, False, bf16=True)
    opt['path']['models'] = str(tmp_path / 'models')
    opt['path']['training_states'] = str(tmp_path / 'states')
    os.makedirs(opt['path']['models']), os.makedirs(opt['path']['training_states'])
    model = build_model(opt)
    lq, gt = global_batch(1, 2)
    model.feed_data({'lq': lq, 'gt': gt})
    model.optimize_parameters(1)
    _raise_word(cuda)
    with pytest.raises(_lib.SrHipError, match='timed out'):
        model.save(0, 1)
    assert os.listdir(opt['path']['models']) == [] and os.listdir(opt['path']['training_states']) == []
    model.recover_from_timeout()
    _lib.check(_lib.load().sr_set_conv_chain(3), 'sr_set_conv_chain')
    model.save(0, 1)
    assert sorted(os.listdir(opt['path']['models'])) == ['net_d_1.pth', 'net_g_1.pth']


@pytest.mark.parametrize('mt', ['SRModel', 'ESRGANModel'])
def test_optimiser_and_ema_refuse_a_timed_out_step_on_the_device_and_training_goes_on(cuda, mt):
    """The window the host cannot close: the device knows a launch of this step timed out (sr_abort_latch is up), the host has not
    seen the copied word yet and issues Adam and the EMA blend.  Both kernels must leave parameters, moments and the EMA shadow
    bitwise unchanged and count the refusal; after recover_from_timeout() the repeated iteration gives exactly what an undisturbed
    run gives (reference contract: base_model.py:78-83 optimiser state, :50-57 EMA)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'helpers'))
    from dp_worker import global_batch, options
    from image_restoration_amd.models import build_model
    from image_restoration_amd.utils.options import set_random_seed
    lib = _lib.load()
    lib.sr_dev_abort_latch_from.argtypes = [C.c_void_p, C.c_void_p]
    lib.sr_dev_abort_latch_from.restype = None
    _lib.check(lib.sr_set_conv_chain(2), 'sr_set_conv_chain')   # both runs on the launch the recovery switches to
    lq, gt = global_batch(1, 2)

    def build():
        set_random_seed(7)
        opt = options(mt, 0, 1, False, bf16=True)
        opt['train']['ema_decay'] = 0.9
        return build_model(opt)

    def state(m):
        out = {}
        for label, pack in m.packs.items():
            out[label + '.p'] = pack.adam.flat_p.clone()
            out[label + '.m'] = pack.adam.exp_avg.clone()
            out[label + '.v'] = pack.adam.exp_avg_sq.clone()
            if pack.shadow_arena is not None:
                out[label + '.ema'] = pack.shadow_arena.clone()
            for name, buf in pack.net.named_buffers():   # BatchNorm running statistics of the discriminator
                out[label + '.buf.' + name] = buf.clone()
        return out

    def step(m, i):
        m.feed_data({'lq': lq, 'gt': gt})
        m.optimize_parameters(i)

    clean, hit = build(), build()
    for i in (1, 2, 3):
        step(clean, i)
    step(hit, 1)
    before = state(hit)
    counts = {k: p.adam.step_count for k, p in hit.packs.items()}
    word = torch.ones(4, dtype=torch.int32, device=cuda)
    lib.sr_dev_abort_latch_from(word.data_ptr(), torch.cuda.current_stream().cuda_stream)   # device only: the host sees nothing
    step(hit, 2)                        # issued in full, refused on the device
    torch.cuda.synchronize()
    after = state(hit)
    for k in before:
        assert torch.equal(before[k], after[k]), k
    assert all(int(p.adam.skipped.item()) == 1 for p in hit.packs.values())
    refused = hit.recover_from_timeout()
    assert refused == {k: 1 for k in hit.packs} and {k: p.adam.step_count for k, p in hit.packs.items()} == counts
    step(hit, 2)
    step(hit, 3)
    torch.cuda.synchronize()
    want, got = state(clean), state(hit)
    for k in want:
        assert torch.equal(want[k], got[k]), k
    assert hit.get_current_log() == clean.get_current_log()

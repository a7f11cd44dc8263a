"""What every SR training model shares on the MI355X path.

Behavioural counterpart of basicsr/models/base_model.py — the contract is the *behaviour* a caller of the reference
sees: device choice (:18), EMA (:50-57), Adam only (:78-83), the two restart schedulers (:85-96), linear warm-up
(:144-168), the `net_<label>_<iter>.pth` / `<iter>.state` file formats and key names (:170-326) and the loss vector
averaged onto rank 0 (:328-353).  The structure is this build's own: networks are held as ``NetPack``s (module +
flat arenas + fused Adam + EMA shadow, models/netpack.py) instead of DDP-wrapped modules with per-tensor optimiser
state.

Data parallelism (reference: DistributedDataParallel, base_model.py:62-76).  There is no wrapper module.  Three
things reproduce what DDP does for the reference:

1. ``align_replicas()`` at the end of construction — parameters, buffers (BatchNorm statistics and counters,
   spectral-norm u/v) and the EMA shadow of every pack become rank 0's.  This is DDP's constructor broadcast;
   per-rank seeding stays ``manual_seed + rank`` as in the reference (options.py:148), so data order and
   augmentation differ per rank while the networks do not.
2. ``NetPack.update`` — one SUM all-reduce of the gradient arena per optimiser step, mean folded into Adam.
3. ``refresh_buffers()`` at the start of every step when ``dist_params.broadcast_buffers`` is not false — DDP's
   ``broadcast_buffers=True`` re-sends rank 0's buffers before each forward.  Train-mode BatchNorm never reads its
   running statistics and only rank 0 saves / validates, so once per step (instead of once per forward) is
   indistinguishable from outside; the deviation is stated here for completeness.
"""
import logging
import os
import time
from collections import OrderedDict

import torch

from ..utils.dist_util import master_only
from . import lr_scheduler
from .netpack import NetPack

_SCHEDULES = {'MultiStepLR': lr_scheduler.MultiStepRestartLR, 'MultiStepRestartLR': lr_scheduler.MultiStepRestartLR,
              'CosineAnnealingRestartLR': lr_scheduler.CosineAnnealingRestartLR}


def _bare(net):
    """A module another framework wrapped (``.module``) is unwrapped; ours never are."""
    inner = getattr(net, 'module', None)
    return inner if isinstance(inner, torch.nn.Module) else net


def _strip_module_prefix(state):
    return OrderedDict((k[len('module.'):] if k.startswith('module.') else k, v) for k, v in state.items())


def _write_with_retries(payload, path, what, attempts=3):
    log = logging.getLogger('basicsr')
    for left in range(attempts - 1, -1, -1):
        try:
            torch.save(payload, path)
            return
        except Exception as exc:  # noqa: BLE001 - a full disk or a flaky mount: wait and try again
            log.warning(f'Save {what} error: {exc}, remaining retry times: {left}')
            time.sleep(1)
    raise IOError(f'Cannot save {path}.')


class BaseModel:

    def __init__(self, opt):
        self.opt = opt
        self.is_train = opt['is_train']
        self.device = torch.device('cpu' if opt['num_gpu'] == 0 else 'cuda')
        self.distributed = bool(opt.get('dist'))
        self.logger = logging.getLogger('basicsr')
        if not self.distributed and isinstance(opt['num_gpu'], int) and opt['num_gpu'] > 1:
            # the reference falls back to single-process nn.DataParallel here (base_model.py:74-75); this build scales only as
            # one process per GPU, so say what happens instead of silently using one device
            self.logger.warning(
                f"num_gpu = {opt['num_gpu']} without a launcher: there is no single-process DataParallel on this path, the job "
                f"runs on cuda:{torch.cuda.current_device() if torch.cuda.is_available() else 0} only.  For {opt['num_gpu']} GPUs "
                f"start it as `python -m torch.distributed.run --nproc-per-node {opt['num_gpu']} --master-addr 127.0.0.1 -m "
                'image_restoration_amd.train -opt <yml> --launcher pytorch`.')
        self.packs = OrderedDict()   # label ('g', 'd') -> NetPack
        self.schedulers = []
        self.log_dict = OrderedDict()
        self._log_staged = None   # (names, device vector) of the last step, not yet read back (stage_loss_dict)

    # ------------------------------------------------------------------ interface of the pipelines
    def feed_data(self, data):
        raise NotImplementedError

    def optimize_parameters(self, current_iter):
        raise NotImplementedError

    def save(self, epoch, current_iter):
        raise NotImplementedError

    def get_current_log(self):
        """The last step's losses as python floats (base_model.py:328-353 fills them every step with ``.item()``: one host
        synchronisation per step, which here would drain the launch queue between steps — the read-back happens when somebody asks)."""
        if self._log_staged is not None:
            names, vec = self._log_staged
            self._log_staged = None
            self.log_dict = OrderedDict(zip(names, vec.tolist()))   # the synchronisation: the step's launches have finished
            if self.device.type == 'cuda':
                from .. import watchdog
                watchdog.verify('get_current_log', synchronize=False)   # numbers of a step that timed out must not reach the log
        return self.log_dict

    def recover_from_timeout(self):
        """After ``optimize_parameters`` / ``get_current_log`` / ``save`` raised the watchdog's SrHipError (a fused dense-block
        launch timed out: a shared or partitioned GPU).  The optimiser and EMA kernels of the affected steps refused to run ON THE
        DEVICE (sr_abort_latch), so parameters, Adam moments and the EMA shadow are those of the last good step.  This puts the
        host side back in line — step counts of the optimisers, the device latch, the watchdog's record — switches this process
        to the chain launch (no co-residency needed, same bits) and returns the number of optimiser steps that were refused per
        network.  The caller repeats those iterations; no checkpoint reload is needed."""
        from .. import _lib, watchdog
        lib = _lib.load()
        torch.cuda.synchronize()
        refused = {label: pack.adam.take_back_skipped() for label, pack in self.packs.items() if pack.adam is not None}
        _lib.check(lib.sr_abort_latch_clear(torch.cuda.current_stream().cuda_stream), 'sr_abort_latch_clear')
        torch.cuda.synchronize()
        lib.sr_chain_watchdog()            # the record of the time-out that was just handled
        _lib.check(lib.sr_set_conv_chain(2), 'sr_set_conv_chain')
        self._log_staged = None
        self.logger.warning(f'recovered from a dense-block time-out: optimiser steps refused on the device {refused}; this process '
                            'uses the chain launch from now on')
        return refused

    # ------------------------------------------------------------------ networks
    def adopt(self, label, net, weights_key=None, shadow_key=None):
        """Puts ``net`` on the device, loads ``path.pretrain_network_<label>`` when given and registers the pack.
        ``path.strict_load_<label>`` and the ``params`` / ``params_ema`` file keys are the reference's."""
        net = net.to(self.device)
        self.report_network(net)
        source = self.opt['path'].get(f'pretrain_network_{label}')
        if source is not None:
            self.load_network(net, source, self.opt['path'].get(f'strict_load_{label}', True),
                              weights_key or 'params')
        self.packs[label] = pack = NetPack(label, net)
        return pack

    def model_to_device(self, net):
        """Kept for callers of the reference's name: device move only, never a wrapper."""
        return net.to(self.device)

    def get_bare_model(self, net):
        return _bare(net)

    @property
    def optimizers(self):
        return [p.adam for p in self.packs.values() if p.adam is not None]

    @master_only
    def report_network(self, net):
        net = _bare(net)
        count = sum(p.numel() for p in net.parameters())
        self.logger.info(f'Network: {type(net).__name__}, with parameters: {count:,d}')
        self.logger.info(str(net))

    print_network = report_network

    def align_replicas(self):
        """DDP's constructor-time ``_sync_module_states`` for every pack (module docstring, point 1)."""
        if self.distributed and self.opt.get('world_size', 1) > 1:
            for pack in self.packs.values():
                pack.align_replicas()

    def refresh_buffers(self):
        if self.distributed and self.opt.get('world_size', 1) > 1 and \
                (self.opt.get('dist_params') or {}).get('broadcast_buffers', True):
            for pack in self.packs.values():
                if any(True for _ in pack.net.buffers()):
                    pack.align_replicas(buffers_only=True)

    def model_ema(self, decay=0.999):
        self.packs['g'].blend_shadow(decay)

    # ------------------------------------------------------------------ optimisers / schedules
    def make_adam(self, pack, block):
        block = dict(block)
        kind = block.pop('type')
        if kind != 'Adam':
            raise NotImplementedError(f'optimizer {kind} is not supperted yet.')
        return pack.attach_adam(block, self.logger)

    def setup_schedulers(self):
        block = dict(self.opt['train']['scheduler'])
        kind = block.pop('type')
        if kind not in _SCHEDULES:
            raise NotImplementedError(f'Scheduler {kind} is not implemented yet.')
        self.schedulers = [_SCHEDULES[kind](adam, **block) for adam in self.optimizers]

    def update_learning_rate(self, current_iter, warmup_iter=-1):
        """Schedulers advance from iteration 2 on; below ``warmup_iter`` the rate is initial_lr * iter / warmup_iter."""
        if current_iter > 1:
            for s in self.schedulers:
                s.step()
        if current_iter < warmup_iter:
            ramp = current_iter / warmup_iter
            for adam in self.optimizers:
                for group in adam.param_groups:
                    group['lr'] = group['initial_lr'] * ramp

    def get_current_learning_rate(self):
        return [group['lr'] for group in self.optimizers[0].param_groups]

    # ------------------------------------------------------------------ files (formats of base_model.py:170-326)
    @master_only
    def save_network(self, net, net_label, current_iter, param_key='params'):
        nets = net if isinstance(net, list) else [net]
        keys = param_key if isinstance(param_key, list) else [param_key]
        assert len(nets) == len(keys), 'The lengths of net and param_key should be the same.'
        tag = 'latest' if current_iter == -1 else current_iter
        payload = {key: _strip_module_prefix(OrderedDict((k, v.detach().cpu().clone())
                                                         for k, v in _bare(n).state_dict().items()))
                   for n, key in zip(nets, keys)}
        _write_with_retries(payload, os.path.join(self.opt['path']['models'], f'{net_label}_{tag}.pth'), 'model')

    def load_network(self, net, load_path, strict=True, param_key='params'):
        net = _bare(net)
        self.logger.info(f'Loading {type(net).__name__} model from {load_path}.')
        blob = torch.load(load_path, map_location='cpu', weights_only=False)
        if param_key is not None:
            if param_key not in blob and 'params' in blob:
                self.logger.info('Loading: params_ema does not exist, use params.')
                param_key = 'params'
            blob = blob[param_key]
        blob = _strip_module_prefix(blob)
        if not strict:
            have = net.state_dict()
            for k in [k for k in blob if k in have and have[k].shape != blob[k].shape]:
                self.logger.warning(f'Size different, ignore [{k}]: crt_net: {have[k].shape}; load_net: {blob[k].shape}')
                blob[k + '.ignore'] = blob.pop(k)
        net.load_state_dict(blob, strict=strict)
        if hasattr(net, 'invalidate_packed'):
            net.invalidate_packed()
        from .. import hip_ops
        hip_ops.invalidate_packs()   # every cached weight image of the process (per-layer caches key on identity + version + epoch)

    @master_only
    def save_training_state(self, epoch, current_iter):
        if current_iter == -1:
            return
        state = dict(epoch=epoch, iter=current_iter, optimizers=[a.state_dict() for a in self.optimizers],
                     schedulers=[s.state_dict() for s in self.schedulers])
        _write_with_retries(state, os.path.join(self.opt['path']['training_states'], f'{current_iter}.state'),
                            'training state')

    def resume_training(self, resume_state):
        adam_states, schedule_states = resume_state['optimizers'], resume_state['schedulers']
        assert len(adam_states) == len(self.optimizers), 'Wrong lengths of optimizers'
        assert len(schedule_states) == len(self.schedulers), 'Wrong lengths of schedulers'
        for adam, st in zip(self.optimizers, adam_states):
            adam.load_state_dict(st)
        for sched, st in zip(self.schedulers, schedule_states):
            sched.load_state_dict(st)

    # ------------------------------------------------------------------ logging
    def reduce_loss_dict(self, loss_dict):
        """name -> python float; under data parallelism rank 0 receives the mean over ranks (dist.reduce then
        / world_size), other ranks keep whatever the reduce left them, as in the reference."""
        names = list(loss_dict)
        if not names:
            return OrderedDict()
        with torch.no_grad():
            vec = torch.stack([loss_dict[n].detach().float().mean() for n in names])
            if self.distributed:
                torch.distributed.reduce(vec, dst=0)
                if self.opt['rank'] == 0:
                    vec /= self.opt['world_size']
            return OrderedDict(zip(names, vec.tolist()))

    def stage_loss_dict(self, loss_dict):
        """reduce_loss_dict without the read-back: the (rank-reduced) loss vector stays on the device until get_current_log()."""
        names = list(loss_dict)
        if not names:
            self._log_staged, self.log_dict = None, OrderedDict()
            return
        with torch.no_grad():
            vec = torch.stack([loss_dict[n].detach().float().mean() for n in names])
            if self.distributed:
                torch.distributed.reduce(vec, dst=0)
                if self.opt['rank'] == 0:
                    vec /= self.opt['world_size']
        self._log_staged = (names, vec)

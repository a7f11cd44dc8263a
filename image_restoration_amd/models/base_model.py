"""Training-side plumbing shared by the SR models.

Counterpart of basicsr/models/base_model.py: device choice (:18), EMA (:50-57), optimiser factory (:78-83, Adam
only), schedulers (:85-96), warm-up / learning-rate update (:144-168), network and training-state save/load
(:170-326, same file formats and key names) and loss reduction to rank 0 (:328-353).

MI355X differences: networks are not wrapped in DistributedDataParallel — every network's parameters and
gradients live in flat arenas (optim.FlatAdam) and data parallelism is one RCCL all-reduce of the gradient
arena per optimiser step (reference semantics: DDP gradient mean, base_model.py:70-73).
"""
import logging
import os
import time
from collections import OrderedDict
from copy import deepcopy

import torch

from .. import optim
from ..utils.dist_util import master_only
from . import lr_scheduler


class BaseModel:

    def __init__(self, opt):
        self.opt = opt
        self.device = torch.device('cuda' if opt['num_gpu'] != 0 else 'cpu')
        self.is_train = opt['is_train']
        self.schedulers = []
        self.optimizers = []
        self.logger = logging.getLogger('basicsr')

    def feed_data(self, data):
        pass

    def optimize_parameters(self):
        pass

    def save(self, epoch, current_iter):
        pass

    def get_current_log(self):
        return self.log_dict

    # ------------------------------------------------------------------ networks
    def model_to_device(self, net):
        """Moves the network to the device.  No DDP/DataParallel wrapper: see the module docstring."""
        return net.to(self.device)

    def get_bare_model(self, net):
        return net.module if hasattr(net, 'module') and isinstance(net.module, torch.nn.Module) else net

    @master_only
    def print_network(self, net):
        net = self.get_bare_model(net)
        n = sum(p.numel() for p in net.parameters())
        self.logger.info(f'Network: {net.__class__.__name__}, with parameters: {n:,d}')
        self.logger.info(str(net))

    def model_ema(self, decay=0.999):
        """net_g_ema = decay*net_g_ema + (1-decay)*net_g over the parameter arenas (one launch)."""
        if decay == 0 or not self._ema_flat.is_cuda:
            with torch.no_grad():
                if decay == 0:
                    self._ema_flat.copy_(self.optimizer_g.flat_p)
                else:
                    raise RuntimeError('EMA update runs only on a HIP device')
            self.net_g_ema.invalidate_packed()
            return
        optim.ema_update(self._ema_flat, self.optimizer_g.flat_p, decay, modules=[self.net_g_ema])

    # ------------------------------------------------------------------ optimisers / schedules
    def get_optimizer(self, optim_type, params, lr, modules=(), **kwargs):
        if optim_type == 'Adam':
            return optim.FlatAdam(params, lr, modules=modules, **kwargs)
        raise NotImplementedError(f'optimizer {optim_type} is not supperted yet.')

    def setup_schedulers(self):
        train_opt = self.opt['train']
        scheduler_type = train_opt['scheduler'].pop('type')
        if scheduler_type in ['MultiStepLR', 'MultiStepRestartLR']:
            cls = lr_scheduler.MultiStepRestartLR
        elif scheduler_type == 'CosineAnnealingRestartLR':
            cls = lr_scheduler.CosineAnnealingRestartLR
        else:
            raise NotImplementedError(f'Scheduler {scheduler_type} is not implemented yet.')
        for optimizer in self.optimizers:
            self.schedulers.append(cls(optimizer, **train_opt['scheduler']))

    def _set_lr(self, lr_groups_l):
        for optimizer, lr_groups in zip(self.optimizers, lr_groups_l):
            for param_group, lr in zip(optimizer.param_groups, lr_groups):
                param_group['lr'] = lr

    def _get_init_lr(self):
        return [[v['initial_lr'] for v in optimizer.param_groups] for optimizer in self.optimizers]

    def update_learning_rate(self, current_iter, warmup_iter=-1):
        if current_iter > 1:
            for scheduler in self.schedulers:
                scheduler.step()
        if current_iter < warmup_iter:  # linear warm-up
            init_lr_g_l = self._get_init_lr()
            self._set_lr([[v / warmup_iter * current_iter for v in init_lr_g] for init_lr_g in init_lr_g_l])

    def get_current_learning_rate(self):
        return [param_group['lr'] for param_group in self.optimizers[0].param_groups]

    # ------------------------------------------------------------------ checkpoints (reference file formats)
    @staticmethod
    def _save_with_retry(obj, path, what):
        retry = 3
        while retry > 0:
            try:
                torch.save(obj, path)
            except Exception as e:  # noqa: BLE001
                logging.getLogger('basicsr').warning(f'Save {what} error: {e}, remaining retry times: {retry - 1}')
                time.sleep(1)
            else:
                break
            finally:
                retry -= 1
        if retry == 0:
            raise IOError(f'Cannot save {path}.')

    @master_only
    def save_network(self, net, net_label, current_iter, param_key='params'):
        if current_iter == -1:
            current_iter = 'latest'
        save_path = os.path.join(self.opt['path']['models'], f'{net_label}_{current_iter}.pth')
        net = net if isinstance(net, list) else [net]
        param_key = param_key if isinstance(param_key, list) else [param_key]
        assert len(net) == len(param_key), 'The lengths of net and param_key should be the same.'
        save_dict = {}
        for net_, key_ in zip(net, param_key):
            sd = OrderedDict()
            for k, v in self.get_bare_model(net_).state_dict().items():
                sd[k[7:] if k.startswith('module.') else k] = v.detach().cpu().clone()
            save_dict[key_] = sd
        self._save_with_retry(save_dict, save_path, 'model')

    def load_network(self, net, load_path, strict=True, param_key='params'):
        net = self.get_bare_model(net)
        self.logger.info(f'Loading {net.__class__.__name__} model from {load_path}.')
        load_net = torch.load(load_path, map_location='cpu', weights_only=False)
        if param_key is not None:
            if param_key not in load_net and 'params' in load_net:
                param_key = 'params'
                self.logger.info('Loading: params_ema does not exist, use params.')
            load_net = load_net[param_key]
        for k, v in deepcopy(load_net).items():
            if k.startswith('module.'):
                load_net[k[7:]] = v
                load_net.pop(k)
        if not strict:
            crt = net.state_dict()
            for k in set(crt) & set(load_net):
                if crt[k].size() != load_net[k].size():
                    self.logger.warning(f'Size different, ignore [{k}]: crt_net: {crt[k].shape}; load_net: {load_net[k].shape}')
                    load_net[k + '.ignore'] = load_net.pop(k)
        net.load_state_dict(load_net, strict=strict)
        if hasattr(net, 'invalidate_packed'):
            net.invalidate_packed()

    @master_only
    def save_training_state(self, epoch, current_iter):
        if current_iter != -1:
            state = {'epoch': epoch, 'iter': current_iter, 'optimizers': [o.state_dict() for o in self.optimizers],
                     'schedulers': [s.state_dict() for s in self.schedulers]}
            self._save_with_retry(state, os.path.join(self.opt['path']['training_states'], f'{current_iter}.state'),
                                  'training state')

    def resume_training(self, resume_state):
        resume_optimizers = resume_state['optimizers']
        resume_schedulers = resume_state['schedulers']
        assert len(resume_optimizers) == len(self.optimizers), 'Wrong lengths of optimizers'
        assert len(resume_schedulers) == len(self.schedulers), 'Wrong lengths of schedulers'
        for i, o in enumerate(resume_optimizers):
            self.optimizers[i].load_state_dict(o)
        for i, s in enumerate(resume_schedulers):
            self.schedulers[i].load_state_dict(s)

    # ------------------------------------------------------------------ logging
    def reduce_loss_dict(self, loss_dict):
        """Averages the logged scalars over ranks onto rank 0 (dist.reduce + /world_size, base_model.py:336-347)."""
        with torch.no_grad():
            if self.opt['dist']:
                keys = list(loss_dict.keys())
                losses = torch.stack([loss_dict[k].detach().float().reshape(()) for k in keys], 0)
                torch.distributed.reduce(losses, dst=0)
                if self.opt['rank'] == 0:
                    losses /= self.opt['world_size']
                loss_dict = {key: loss for key, loss in zip(keys, losses)}
            log_dict = OrderedDict()
            for name, value in loss_dict.items():
                log_dict[name] = value.mean().item()
            return log_dict

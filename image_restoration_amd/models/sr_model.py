"""SRModel: generator-only training / testing (PSNR pre-training of RRDBNet).

Counterpart of basicsr/models/sr_model.py:14-133,204-209 on the HIP path: same option keys, same step order
(zero_grad -> forward -> pixel loss -> backward -> Adam -> EMA), same log keys."""
from collections import OrderedDict

import os

import torch

from .. import optim
from ..archs import build_network
from ..losses import build_loss
from ..utils.registry import MODEL_REGISTRY
from .base_model import BaseModel


@MODEL_REGISTRY.register()
class SRModel(BaseModel):

    def __init__(self, opt):
        super().__init__(opt)
        self.net_g = self.model_to_device(build_network(opt['network_g']))
        self.print_network(self.net_g)
        load_path = self.opt['path'].get('pretrain_network_g', None)
        if load_path is not None:
            self.load_network(self.net_g, load_path, self.opt['path'].get('strict_load_g', True))
        if self.is_train:
            self.init_training_settings()

    def _init_ema(self, train_opt):
        self.ema_decay = train_opt.get('ema_decay', 0)
        if self.ema_decay > 0:
            self.logger.info(f'Use Exponential Moving Average with decay: {self.ema_decay}')
            self.net_g_ema = build_network(self.opt['network_g']).to(self.device)
            load_path = self.opt['path'].get('pretrain_network_g', None)
            if load_path is not None:
                self.load_network(self.net_g_ema, load_path, self.opt['path'].get('strict_load_g', True), 'params_ema')
            self.net_g_ema.eval()
            self._ema_pending_copy = load_path is None

    def _finish_ema(self):
        """After the optimiser built net_g's arena: flatten the EMA copy the same way."""
        if self.ema_decay > 0:
            self._ema_flat = optim.flatten_parameters(self.net_g_ema)
            if self._ema_pending_copy:
                self.model_ema(0)  # copy net_g weight

    def init_training_settings(self):
        self.net_g.train()
        train_opt = self.opt['train']
        self._init_ema(train_opt)
        self.cri_pix = build_loss(train_opt['pixel_opt']).to(self.device) if train_opt.get('pixel_opt') else None
        # perceptual loss (losses.py:249-356 on HIP VGG features; the frozen VGG is not optimised)
        self.cri_perceptual = (build_loss(train_opt['perceptual_opt']).to(self.device)
                               if train_opt.get('perceptual_opt') else None)
        if self.cri_pix is None and self.cri_perceptual is None:
            raise ValueError('Both pixel and perceptual losses are None.')
        self.setup_optimizers()
        self.setup_schedulers()
        self._finish_ema()

    def setup_optimizers(self):
        train_opt = self.opt['train']
        optim_params = []
        for k, v in self.net_g.named_parameters():
            if v.requires_grad:
                optim_params.append(v)
            else:
                self.logger.warning(f'Params {k} will not be optimized.')
        optim_type = train_opt['optim_g'].pop('type')
        self.optimizer_g = self.get_optimizer(optim_type, optim_params, modules=[self.net_g], **train_opt['optim_g'])
        self.optimizers.append(self.optimizer_g)

    def feed_data(self, data):
        self.lq = data['lq'].to(self.device)
        if 'gt' in data:
            self.gt = data['gt'].to(self.device)

    def _step(self, optimizer):
        """DP gradient exchange (one arena all-reduce) + fused Adam."""
        scale = optimizer.all_reduce_grads() if self.opt['dist'] else 1.0
        optimizer.step(grad_scale=scale)

    def optimize_parameters(self, current_iter):
        self.optimizer_g.zero_grad()
        self.output = self.net_g(self.lq)
        loss_dict = OrderedDict()
        l_total = 0
        if self.cri_pix:  # sr_model.py:97-108
            l_pix = self.cri_pix(self.output, self.gt)
            l_total = l_total + l_pix
            loss_dict['l_pix'] = l_pix
        if self.cri_perceptual:
            l_percep, l_style = self.cri_perceptual(self.output, self.gt)
            if l_percep is not None:
                l_total = l_total + l_percep
                loss_dict['l_percep'] = l_percep
            if l_style is not None:
                l_total = l_total + l_style
                loss_dict['l_style'] = l_style
        l_total.backward()
        self._step(self.optimizer_g)
        self.log_dict = self.reduce_loss_dict(loss_dict)
        if self.ema_decay > 0:
            self.model_ema(decay=self.ema_decay)

    def test(self):
        if hasattr(self, 'net_g_ema'):
            self.net_g_ema.eval()
            with torch.no_grad():
                self.output = self.net_g_ema(self.lq)
        else:
            self.net_g.eval()
            with torch.no_grad():
                self.output = self.net_g(self.lq)
            self.net_g.train()

    # ------------------------------------------------------------------ validation (sr_model.py:131-184)
    def validation(self, dataloader, current_iter, tb_logger=None, save_img=False):
        """base_model.py:39-48: rank 0 validates when distributed."""
        if self.opt['dist']:
            if self.opt['rank'] == 0:
                self.nondist_validation(dataloader, current_iter, tb_logger, save_img)
        else:
            self.nondist_validation(dataloader, current_iter, tb_logger, save_img)

    def nondist_validation(self, dataloader, current_iter, tb_logger=None, save_img=False):
        """The reference's loop (sr_model.py:135-184): per validation image feed_data -> test -> metrics, averaged over
        the loader and logged; metric options come from opt['val']['metrics'] ({name: {type: calculate_psnr |
        calculate_ssim, crop_border, test_y_channel}}).  PSNR / SSIM without test_y_channel are reduced on the device
        from the fp32 output with tensor2img's quantisation (metrics/psnr.py: no device->host image copy); other
        metric options take the reference's host route through tensor2img."""
        import os.path as osp
        from ..metrics import psnr_device, ssim_device
        from ..utils.img_util import tensor2img
        from ..utils.registry import METRIC_REGISTRY
        dataset_name = dataloader.dataset.opt['name'] if hasattr(dataloader.dataset, 'opt') else 'val'
        val_opt = self.opt.get('val') or {}
        metrics = val_opt.get('metrics')
        if metrics is not None:
            self.metric_results = {m: 0 for m in metrics.keys()}
        idx, scored = -1, 0
        for idx, val_data in enumerate(dataloader):
            self.feed_data(val_data)
            self.test()
            out, gt = self.output.detach(), getattr(self, 'gt', None)
            if save_img:
                import numpy as np
                from PIL import Image
                name = osp.splitext(osp.basename(val_data['lq_path'][0]))[0] if 'lq_path' in val_data else f'{idx:06d}'
                vis = self.opt['path'].get('visualization', '.')
                path = osp.join(vis, name, f'{name}_{current_iter}.png') if self.opt.get('is_train', True) else \
                    osp.join(vis, dataset_name, f'{name}_{val_opt.get("suffix") or self.opt["name"]}.png')
                os.makedirs(osp.dirname(path), exist_ok=True)
                Image.fromarray(np.ascontiguousarray(tensor2img([out[0:1].cpu()], rgb2bgr=False))).save(path)
            if metrics is not None and gt is None and idx == 0:
                self.logger.warning(f'{dataset_name} has no ground truth: images only, no metrics.')
            if metrics is not None and gt is not None:
                scored += 1
                for mname, mopt in metrics.items():
                    mopt = dict(mopt)
                    mtype = mopt.pop('type')
                    crop = mopt.get('crop_border', 0)
                    if mtype in ('calculate_psnr', 'calculate_ssim') and not mopt.get('test_y_channel', False):
                        fn = psnr_device if mtype == 'calculate_psnr' else ssim_device
                        self.metric_results[mname] += fn(out[0:1], gt[0:1], crop)[0]  # first image, like the reference
                    else:
                        sr_img, gt_img = tensor2img([out[0:1].cpu()]), tensor2img([gt[0:1].cpu()])
                        self.metric_results[mname] += METRIC_REGISTRY.get(mtype)(sr_img, gt_img, **mopt)
            if hasattr(self, 'gt'):
                del self.gt
            del self.lq, self.output
        if metrics is not None and scored > 0:
            for m in self.metric_results:
                self.metric_results[m] /= scored
            log = f'Validation {dataset_name}\n' + ''.join(f'\t # {m}: {v:.4f}\n' for m, v in self.metric_results.items())
            self.logger.info(log)
            if tb_logger:
                for m, v in self.metric_results.items():
                    tb_logger.add_scalar(f'metrics/{m}', v, current_iter)

    def get_current_visuals(self):
        out_dict = OrderedDict()
        out_dict['lq'] = self.lq.detach().cpu()
        out_dict['result'] = self.output.detach().cpu()
        if hasattr(self, 'gt'):
            out_dict['gt'] = self.gt.detach().cpu()
        return out_dict

    def save(self, epoch, current_iter):
        if hasattr(self, 'net_g_ema'):
            self.save_network([self.net_g, self.net_g_ema], 'net_g', current_iter, param_key=['params', 'params_ema'])
        else:
            self.save_network(self.net_g, 'net_g', current_iter)
        self.save_training_state(epoch, current_iter)

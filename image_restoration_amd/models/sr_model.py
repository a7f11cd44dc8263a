"""SRModel: generator-only training / testing (the PSNR pre-training stage of RRDBNet).

Behavioural counterpart of basicsr/models/sr_model.py:14-133,204-209: same option keys (``network_g``,
``path.pretrain_network_g / strict_load_g``, ``train.{ema_decay, pixel_opt, perceptual_opt, optim_g, scheduler}``),
same step (zero grads -> forward -> content loss -> backward -> Adam -> EMA), same log keys (``l_pix``, ``l_percep``,
``l_style``), same checkpoint files.  Structure: the generator is a ``NetPack``; the terms of a loss are collected in a
``LossBook`` that GAN models extend with their own entries.
"""
import os
from collections import OrderedDict

import torch

from .. import watchdog
from ..archs import build_network
from ..losses import build_loss
from ..utils.registry import MODEL_REGISTRY
from .base_model import BaseModel


class LossBook:
    """Ordered ledger of the scalars of one step: every entry is logged, entries flagged ``optimise`` add up to the
    objective that is back-propagated."""

    def __init__(self):
        self.entries = OrderedDict()
        self.objective = None

    def log(self, key, value):
        self.entries[key] = value

    def charge(self, key, value):
        self.entries[key] = value
        self.objective = value if self.objective is None else self.objective + value


@MODEL_REGISTRY.register()
class SRModel(BaseModel):

    log_prefix = 'l_'   # SRGAN / ESRGAN log the generator's content terms as l_g_*

    def __init__(self, opt):
        super().__init__(opt)
        self.gen = self.adopt('g', build_network(opt['network_g']))
        self.net_g = self.gen.net
        if self.is_train:
            self.init_training_settings()
        self.align_replicas()

    # ------------------------------------------------------------------ set-up
    def init_training_settings(self):
        cfg = self.opt['train']
        self.net_g.train()
        self._build_content_losses(cfg)
        if self.cri_pix is None and self.cri_perceptual is None:
            raise ValueError('Both pixel and perceptual losses are None.')
        self.setup_optimizers()
        self.setup_schedulers()
        self._build_shadow(cfg)

    def _build_content_losses(self, cfg):
        def criterion(key):
            return build_loss(cfg[key]).to(self.device) if cfg.get(key) else None
        self.cri_pix = criterion('pixel_opt')
        self.cri_perceptual = criterion('perceptual_opt')   # frozen VGG features (losses/perceptual_loss.py)

    def _build_shadow(self, cfg):
        """``train.ema_decay`` > 0: a second generator that trails the trained one.  It starts from the checkpoint's
        ``params_ema`` when a pretrained file is given, else as a copy of the live weights (sr_model.py:38-50)."""
        self.ema_decay = cfg.get('ema_decay', 0)
        if not self.ema_decay > 0:
            return
        self.logger.info(f'Use Exponential Moving Average with decay: {self.ema_decay}')
        twin = build_network(self.opt['network_g']).to(self.device)
        source = self.opt['path'].get('pretrain_network_g')
        if source is not None:
            self.load_network(twin, source, self.opt['path'].get('strict_load_g', True), 'params_ema')
        self.gen.attach_shadow(twin)
        self.net_g_ema = twin
        if source is None:
            self.gen.blend_shadow(0)

    def setup_optimizers(self):
        self.optimizer_g = self.make_adam(self.gen, self.opt['train']['optim_g'])

    # ------------------------------------------------------------------ one step
    def feed_data(self, data):
        for key in ('lq', 'gt'):
            if key in data:
                setattr(self, key, data[key].to(self.device))

    def content_terms(self, book):
        """Pixel / perceptual / style terms of the generator objective, in the reference's order."""
        pre = self.log_prefix
        if self.cri_pix:
            book.charge(pre + 'pix', self.cri_pix(self.output, self.gt))
        if self.cri_perceptual:
            percep, style = self.cri_perceptual(self.output, self.gt)
            if percep is not None:
                book.charge(pre + 'percep', percep)
            if style is not None:
                book.charge(pre + 'style', style)

    def finish_step(self, book):
        self.stage_loss_dict(book.entries)   # read back by get_current_log(): no host synchronisation inside a step
        if self.device.type == 'cuda':
            # (without a synchronisation: reports what the abort words copied behind EARLIER launches say — a time-out surfaces
            # a step or two later, and at every hand-over: get_current_log, save, test.  The state stays valid meanwhile: the Adam
            # and EMA kernels of a step whose launches timed out see the device latch and refuse — BaseModel.recover_from_timeout)
            watchdog.verify('optimize_parameters', synchronize=False)
        if self.ema_decay > 0:
            self.gen.blend_shadow(self.ema_decay)

    def optimize_parameters(self, current_iter):
        self.refresh_buffers()
        book = LossBook()
        self.gen.clear_grads()
        self.output = self.net_g(self.lq)
        self.content_terms(book)
        book.objective.backward()
        self.gen.update(self.distributed)
        self.finish_step(book)

    # ------------------------------------------------------------------ inference / validation
    def test(self):
        """One forward without autograd, through the EMA shadow when there is one (sr_model.py:120-129)."""
        runner = getattr(self, 'net_g_ema', None)
        restore = runner is None
        if restore:
            runner = self.net_g
        runner.eval()
        with torch.no_grad():
            if self.lq.is_cuda:   # the image goes to metrics / a PNG next: never hand over a timed-out forward
                self.output = watchdog.guarded(lambda: runner(self.lq), 'SRModel.test')
            else:
                self.output = runner(self.lq)
        if restore:
            self.net_g.train()

    def validation(self, dataloader, current_iter, tb_logger=None, save_img=False):
        """Rank 0 validates alone under data parallelism (base_model.py:39-48)."""
        if not self.distributed or self.opt['rank'] == 0:
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
        shown = OrderedDict(lq=self.lq.detach().cpu(), result=self.output.detach().cpu())
        if hasattr(self, 'gt'):
            shown['gt'] = self.gt.detach().cpu()
        return shown

    # ------------------------------------------------------------------ files
    def save(self, epoch, current_iter):
        """net_g_<iter>.pth holds ``params`` (+ ``params_ema`` when the shadow exists), then the training state."""
        if self.device.type == 'cuda':
            watchdog.verify('save')   # never write a checkpoint behind a step that timed out
        if self.gen.shadow is not None:
            self.save_network([self.net_g, self.gen.shadow], 'net_g', current_iter, param_key=['params', 'params_ema'])
        else:
            self.save_network(self.net_g, 'net_g', current_iter)
        self.save_training_state(epoch, current_iter)

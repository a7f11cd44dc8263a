"""PairedImageDataset: LQ / GT image pairs from two folders, a meta-info file or LMDB.

Behavioural counterpart of basicsr/data/paired_image_dataset.py:11-109: the reference's option keys (dataroot_gt, dataroot_lq,
io_backend, filename_tmpl, meta_info_file, gt_size, use_flip, use_rot, mean, std, scale, phase) and item dict (``lq``, ``gt`` CHW RGB
float32 in [0, 1], ``lq_path``, ``gt_path``); in the train phase a random paired window, then a random symmetry.  PIL decodes
instead of cv2 (same pixels for PNG).

Extension ``device_augment: true`` (train phase): the item carries the *uint8* window pair and the drawn symmetry code instead of
finished float tensors; conversion, channel swap, flips, transposition and normalisation then run as ONE HIP kernel per batch on
the copy stream (data/device_pipeline.py).  The random draws are the same calls in the same order, so a seeded run yields the same
training tensors either way — 4x fewer bytes over PCIe and no float work on the host cores."""
import numpy as np
from torch.utils.data import Dataset

from ..utils.registry import DATASET_REGISTRY
from . import data_util, transforms
from .image_source import ImageSource


@DATASET_REGISTRY.register()
class PairedImageDataset(ImageSource, Dataset):

    def __init__(self, opt):
        Dataset.__init__(self)
        self.lq_folder, self.gt_folder = opt['dataroot_lq'], opt['dataroot_gt']
        self._init_source(opt, (self.lq_folder, self.gt_folder), ('lq', 'gt'))
        self.filename_tmpl = opt.get('filename_tmpl', '{}')
        dirs, keys = [self.lq_folder, self.gt_folder], ['lq', 'gt']
        if self.io_backend_opt['type'] == 'lmdb':
            self.paths = data_util.paired_paths_from_lmdb(dirs, keys)
        elif opt.get('meta_info_file') is not None:
            self.paths = data_util.paired_paths_from_meta_info_file(dirs, keys, opt['meta_info_file'], self.filename_tmpl)
        else:
            self.paths = data_util.paired_paths_from_folder(dirs, keys, self.filename_tmpl)
        self.training = opt['phase'] == 'train'
        mode = opt.get('device_augment', False)
        self.device_augment = bool(mode) and self.training
        self.ship_whole_images = self.device_augment and mode == 'full'   # equally sized images: the crop runs on the device too

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        entry = self.paths[index]
        where = {'lq_path': entry['lq_path'], 'gt_path': entry['gt_path']}
        if self.device_augment:
            return dict(self._raw_patch_pair(entry), **where)
        gt = self.decode(entry['gt_path'], 'gt')
        lq = self.decode(entry['lq_path'], 'lq')
        if self.training:
            gt, lq = transforms.paired_random_crop(gt, lq, self.opt['gt_size'], self.opt['scale'], entry['gt_path'])
            gt, lq = transforms.augment([gt, lq], self.opt['use_flip'], self.opt['use_rot'])
        gt, lq = self.to_tensors(gt, lq)
        return dict(lq=lq, gt=gt, **where)

    def _raw_patch_pair(self, entry):
        scale, gt_size = self.opt['scale'], self.opt['gt_size']
        gt = self.decode(entry['gt_path'], 'gt', as_float=False)
        lq = self.decode(entry['lq_path'], 'lq', as_float=False)
        lq_patch = transforms.check_pair_geometry(gt.shape, lq.shape, gt_size, scale, entry['gt_path'])
        top, left = transforms.draw_window(lq.shape[0], lq.shape[1], lq_patch)
        code = transforms.draw_symmetry(self.opt['use_flip'], self.opt['use_rot'])
        item = {'sym': np.int32(code)}
        if self.ship_whole_images:
            item['window'] = np.array([top, left], np.int32)
        else:
            (gt,), (lq,) = transforms.cut_pair([gt], [lq], top, left, lq_patch, scale, gt_size)
        item.update(lq_u8=np.ascontiguousarray(lq), gt_u8=np.ascontiguousarray(gt))
        return item

"""PairedImageDataset: LQ / GT image pairs from folders, a meta-info file or LMDB.

Counterpart of basicsr/data/paired_image_dataset.py:11-109 with the reference's option keys (dataroot_gt, dataroot_lq,
io_backend, filename_tmpl, meta_info_file, gt_size, use_flip, use_rot, mean, std, scale, phase) and return dict
(lq, gt CHW RGB float32 in [0, 1], lq_path, gt_path).  Decoding uses PIL instead of cv2 (identical pixels for PNG)."""
import torch
from torch.utils import data as data

from ..utils.img_util import imfrombytes, img2tensor
from ..utils.registry import DATASET_REGISTRY
from .data_util import paired_paths_from_folder, paired_paths_from_lmdb, paired_paths_from_meta_info_file
from .file_client import FileClient
from .transforms import augment, paired_random_crop


@DATASET_REGISTRY.register()
class PairedImageDataset(data.Dataset):

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.file_client = None
        self.io_backend_opt = dict(opt['io_backend'])
        self.mean = opt.get('mean')
        self.std = opt.get('std')
        self.gt_folder, self.lq_folder = opt['dataroot_gt'], opt['dataroot_lq']
        self.filename_tmpl = opt.get('filename_tmpl', '{}')
        if self.io_backend_opt['type'] == 'lmdb':
            self.io_backend_opt['db_paths'] = [self.lq_folder, self.gt_folder]
            self.io_backend_opt['client_keys'] = ['lq', 'gt']
            self.paths = paired_paths_from_lmdb([self.lq_folder, self.gt_folder], ['lq', 'gt'])
        elif opt.get('meta_info_file') is not None:
            self.paths = paired_paths_from_meta_info_file([self.lq_folder, self.gt_folder], ['lq', 'gt'], opt['meta_info_file'],
                                                          self.filename_tmpl)
        else:
            self.paths = paired_paths_from_folder([self.lq_folder, self.gt_folder], ['lq', 'gt'], self.filename_tmpl)

    def __getitem__(self, index):
        if self.file_client is None:  # created lazily, inside the worker process
            kw = dict(self.io_backend_opt)
            self.file_client = FileClient(kw.pop('type'), **kw)
        scale = self.opt['scale']
        gt_path, lq_path = self.paths[index]['gt_path'], self.paths[index]['lq_path']
        img_gt = imfrombytes(self.file_client.get(gt_path, 'gt'), float32=True)  # HWC, BGR, [0, 1]
        img_lq = imfrombytes(self.file_client.get(lq_path, 'lq'), float32=True)
        if self.opt['phase'] == 'train':
            img_gt, img_lq = paired_random_crop(img_gt, img_lq, self.opt['gt_size'], scale, gt_path)
            img_gt, img_lq = augment([img_gt, img_lq], self.opt['use_flip'], self.opt['use_rot'])
        img_gt, img_lq = img2tensor([img_gt, img_lq], bgr2rgb=True, float32=True)
        if self.mean is not None or self.std is not None:  # torchvision.transforms.functional.normalize, in place
            mean = torch.as_tensor(self.mean if self.mean is not None else [0.] * img_gt.size(0), dtype=torch.float32).view(-1, 1, 1)
            std = torch.as_tensor(self.std if self.std is not None else [1.] * img_gt.size(0), dtype=torch.float32).view(-1, 1, 1)
            img_lq.sub_(mean).div_(std)
            img_gt.sub_(mean).div_(std)
        return {'lq': img_lq, 'gt': img_gt, 'lq_path': lq_path, 'gt_path': gt_path}

    def __len__(self):
        return len(self.paths)

"""SingleImageDataset: LQ images only, for the test pipeline without ground truth (basicsr/data/single_image_dataset.py:10-68;
option keys dataroot_lq, meta_info_file, io_backend, mean, std; items {'lq', 'lq_path'}, CHW RGB float32 in [0, 1])."""
import os.path as osp

import torch
from torch.utils import data as data

from ..utils.img_util import imfrombytes, img2tensor
from ..utils.registry import DATASET_REGISTRY
from .data_util import scandir
from .file_client import FileClient


@DATASET_REGISTRY.register()
class SingleImageDataset(data.Dataset):

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.file_client = None
        self.io_backend_opt = dict(opt['io_backend'])
        self.mean, self.std = opt.get('mean'), opt.get('std')
        self.lq_folder = opt['dataroot_lq']
        if self.io_backend_opt['type'] == 'lmdb':
            self.io_backend_opt['db_paths'] = [self.lq_folder]
            self.io_backend_opt['client_keys'] = ['lq']
            if not self.lq_folder.endswith('.lmdb'):
                raise ValueError(f'Folder {self.lq_folder} should in lmdb format.')
            with open(osp.join(self.lq_folder, 'meta_info.txt')) as fin:
                self.paths = [line.split('.')[0] for line in fin]
        elif opt.get('meta_info_file') is not None:
            with open(opt['meta_info_file'], 'r') as fin:
                self.paths = [osp.join(self.lq_folder, line.strip().split(' ')[0]) for line in fin if line.strip()]
        else:
            self.paths = [osp.join(self.lq_folder, name) for name in scandir(self.lq_folder)]

    def __getitem__(self, index):
        if self.file_client is None:
            kw = dict(self.io_backend_opt)
            self.file_client = FileClient(kw.pop('type'), **kw)
        lq_path = self.paths[index]
        img_lq = img2tensor(imfrombytes(self.file_client.get(lq_path, 'lq'), float32=True), bgr2rgb=True, float32=True)
        if self.mean is not None or self.std is not None:
            c = img_lq.size(0)
            mean = torch.as_tensor(self.mean if self.mean is not None else [0.] * c, dtype=torch.float32).view(-1, 1, 1)
            std = torch.as_tensor(self.std if self.std is not None else [1.] * c, dtype=torch.float32).view(-1, 1, 1)
            img_lq.sub_(mean).div_(std)
        return {'lq': img_lq, 'lq_path': lq_path}

    def __len__(self):
        return len(self.paths)

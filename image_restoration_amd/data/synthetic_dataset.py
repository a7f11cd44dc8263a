"""Synthetic paired LQ/GT crops (uniform noise, numpy PCG64) with the sample dict of PairedImageDataset
(`lq`, `gt`, `lq_path`, `gt_path`; paired_image_dataset.py:98): CHW RGB float32 in [0, 1]."""
import numpy as np
import torch
from torch.utils import data

from ..utils.registry import DATASET_REGISTRY


@DATASET_REGISTRY.register()
class SyntheticPairedDataset(data.Dataset):

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.length = int(opt.get('num_samples', 256))
        self.gt_size = int(opt.get('gt_size', 128))
        self.scale = int(opt.get('scale', 4))
        self.seed = int(opt.get('seed', 0))

    def __len__(self):
        return self.length

    def __getitem__(self, index):
        rng = np.random.default_rng(self.seed * 1000003 + index)
        lq = self.gt_size // self.scale
        return {'lq': torch.from_numpy(rng.random((3, lq, lq), dtype=np.float32)),
                'gt': torch.from_numpy(rng.random((3, self.gt_size, self.gt_size), dtype=np.float32)),
                'lq_path': f'synthetic/{index}', 'gt_path': f'synthetic/{index}'}

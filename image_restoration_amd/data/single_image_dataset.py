"""SingleImageDataset: LQ images only — the test pipeline's input when there is no ground truth.

Behavioural counterpart of basicsr/data/single_image_dataset.py:10-68 (option keys dataroot_lq, meta_info_file, io_backend, mean,
std; items ``{'lq', 'lq_path'}``, CHW RGB float32 in [0, 1])."""
import os

from torch.utils.data import Dataset

from ..utils.registry import DATASET_REGISTRY
from . import data_util
from .image_source import ImageSource


@DATASET_REGISTRY.register()
class SingleImageDataset(ImageSource, Dataset):

    def __init__(self, opt):
        Dataset.__init__(self)
        self.lq_folder = root = opt['dataroot_lq']
        self._init_source(opt, (root,), ('lq',))
        if self.io_backend_opt['type'] == 'lmdb':
            if not root.endswith('.lmdb'):
                raise ValueError(f'Folder {root} should in lmdb format.')
            self.paths = data_util.lmdb_keys(root)
        elif opt.get('meta_info_file') is not None:
            with open(opt['meta_info_file']) as f:
                self.paths = [os.path.join(root, ln.split(' ')[0]) for ln in (raw.strip() for raw in f) if ln]
        else:
            self.paths = [os.path.join(root, name) for name in data_util.scandir(root)]

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, index):
        path = self.paths[index]
        (lq,) = self.to_tensors(self.decode(path, 'lq'))
        return {'lq': lq, 'lq_path': path}

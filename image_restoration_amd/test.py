"""Test entry point, the counterpart of basicsr/test.py:11-43: option file (is_train = False) -> one loader (batch 1, in order)
per datasets.* entry -> build_model -> model.validation(loader, current_iter = experiment name, save_img = val.save_img).
Results go to results/<name>/visualization/<dataset name>/<image>_<suffix or name>.png, metrics (when the dataset has
ground truth and val.metrics is set) to the log.

    python -m image_restoration_amd.test -opt options/test/ESRGAN/test_ESRGAN_x4.yml
"""
import logging
import os

from torch.utils.data import DataLoader

from .models import build_model
from .train import build_dataset
from .utils.options import dict2str, parse_options


def test_pipeline(root_path, argv=None):
    opt = parse_options(root_path, is_train=False, argv=argv)
    if opt['rank'] == 0:
        for key in ('results_root', 'visualization'):
            os.makedirs(opt['path'][key], exist_ok=True)
    logging.basicConfig(level=logging.INFO if opt['rank'] == 0 else logging.ERROR, format='%(asctime)s %(levelname)s: %(message)s')
    logger = logging.getLogger('basicsr')
    logger.info(dict2str(opt))
    loaders = []
    for _, dataset_opt in sorted((opt.get('datasets') or {}).items()):
        test_set = build_dataset(dataset_opt)
        logger.info(f"Number of test images in {dataset_opt['name']}: {len(test_set)}")
        loaders.append(DataLoader(test_set, batch_size=1, shuffle=False, num_workers=0))
    model = build_model(opt)
    save_img = (opt.get('val') or {}).get('save_img', False)
    for loader in loaders:
        logger.info(f"Testing {loader.dataset.opt['name']}...")
        model.validation(loader, current_iter=opt['name'], tb_logger=None, save_img=save_img)
    return model


if __name__ == '__main__':
    test_pipeline(os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir)))

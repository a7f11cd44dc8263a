"""Training entry point (counterpart of basicsr/train.py:91-199): parse options -> dataloader -> build_model ->
iteration loop (update_learning_rate, feed_data, optimize_parameters, logging, checkpoints).

    python -m image_restoration_amd.train -opt options/train/ESRGAN/train_ESRGAN_x4_synthetic.yml
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m image_restoration_amd.train \
        -opt training_config/train_rrdbnet_esrgan_x4_mi355x.yml --launcher pytorch
"""
import logging
import math
import os
import time

import torch
from torch.utils.data import DataLoader

from .data import DeviceFeed, DevicePatchPipeline, EnlargedSampler, HostFeed, PrefetchDataLoader
from .models import build_model
from .utils.options import dict2str, parse_options
from .utils.registry import DATASET_REGISTRY


def build_dataset(dataset_opt):
    opt = dict(dataset_opt)
    return DATASET_REGISTRY.get(opt['type'])(opt)


def create_train_loader(opt):
    dataset_opt = next(v for k, v in opt['datasets'].items() if v['phase'] == 'train')
    train_set = build_dataset(dataset_opt)
    ratio = dataset_opt.get('dataset_enlarge_ratio', 1)
    sampler = EnlargedSampler(train_set, opt['world_size'], opt['rank'], ratio)
    batch = dataset_opt['batch_size_per_gpu']
    common = dict(dataset=train_set, batch_size=batch, shuffle=False, sampler=sampler,
                  num_workers=dataset_opt.get('num_worker_per_gpu', 0), drop_last=True, pin_memory=True)
    if dataset_opt.get('prefetch_mode') == 'cpu':   # the reference's PrefetchDataLoader: collation runs ahead on a thread
        loader = PrefetchDataLoader(num_prefetch_queue=dataset_opt.get('num_prefetch_queue', 1), **common)
    else:
        loader = DataLoader(**common)
    iters_per_epoch = math.ceil(len(train_set) * ratio / (batch * opt['world_size']))
    total_iters = int(opt['train']['total_iter'])
    return loader, sampler, math.ceil(total_iters / iters_per_epoch), total_iters


def create_val_loaders(opt):
    """One loader (batch 1, in order, rank-local) per datasets.* entry whose phase is 'val' (train.py:43-66 of the
    reference builds them next to the train loader)."""
    loaders = []
    for k, v in opt['datasets'].items():
        if v.get('phase', k.split('_')[0]) == 'val':
            loaders.append(DataLoader(build_dataset(v), batch_size=1, shuffle=False, num_workers=0))
    return loaders


def load_resume_state(opt):
    """--auto_resume picks the newest experiments/<name>/training_states/*.state (train.py:68-88) and points the
    pretrain paths at the matching networks (check_resume, misc.py:94-117)."""
    state_path = opt['path'].get('resume_state')
    if opt.get('auto_resume'):
        folder = opt['path']['training_states']
        if os.path.isdir(folder):
            states = [f for f in os.listdir(folder) if f.endswith('.state')]
            if states:
                it = max(int(f.split('.state')[0]) for f in states)
                state_path = os.path.join(folder, f'{it}.state')
                opt['path']['resume_state'] = state_path
    if not state_path:
        return None
    state = torch.load(state_path, map_location='cpu', weights_only=False)
    it = state['iter']
    for net in ('g', 'd'):
        if f'network_{net}' in opt:
            opt['path'][f'pretrain_network_{net}'] = os.path.join(opt['path']['models'], f'net_{net}_{it}.pth')
    return state


def train_pipeline(root_path, argv=None):
    opt = parse_options(root_path, is_train=True, argv=argv)
    logging.basicConfig(level=logging.INFO if opt['rank'] == 0 else logging.ERROR,
                        format='%(asctime)s %(levelname)s: %(message)s')
    logger = logging.getLogger('basicsr')
    resume_state = load_resume_state(opt)
    if opt['rank'] == 0:
        for key in ('models', 'training_states', 'visualization'):
            os.makedirs(opt['path'][key], exist_ok=True)
    logger.info(dict2str(opt))
    loader, sampler, total_epochs, total_iters = create_train_loader(opt)
    val_loaders = create_val_loaders(opt)
    val_freq = (opt.get('val') or {}).get('val_freq')
    model = build_model(opt)
    start_epoch, current_iter = 0, 0
    if resume_state:
        model.resume_training(resume_state)
        start_epoch, current_iter = resume_state['epoch'], resume_state['iter']
        logger.info(f"Resuming training from epoch: {start_epoch}, iter: {current_iter}.")
    print_freq = opt['logger']['print_freq']
    save_freq = opt['logger']['save_checkpoint_freq']
    warmup = opt['train'].get('warmup_iter', -1)
    t_iter = time.time()
    # datasets.train.prefetch_mode: None / 'cpu' iterate the loader, 'cuda' stages the next batch on a copy stream
    train_block = next(v for v in opt['datasets'].values() if v['phase'] == 'train')
    prefetch_mode = train_block.get('prefetch_mode')
    device_side = DevicePatchPipeline(train_block) if train_block.get('device_augment') else None
    if device_side is not None and prefetch_mode != 'cuda':
        raise ValueError("device_augment needs prefetch_mode: cuda (the augmentation kernel runs on the feed's copy stream)")
    if prefetch_mode is None or prefetch_mode == 'cpu':
        prefetcher = HostFeed(loader)
    elif prefetch_mode == 'cuda':
        prefetcher = DeviceFeed(loader, opt, pipeline=device_side)
    else:
        raise ValueError(f"Wrong prefetch_mode {prefetch_mode}. Supported ones are: None, 'cuda', 'cpu'.")
    for epoch in range(start_epoch, total_epochs + 1):
        sampler.set_epoch(epoch)
        prefetcher.reset()
        while (data := prefetcher.next()) is not None:
            current_iter += 1
            if current_iter > total_iters:
                break
            model.update_learning_rate(current_iter, warmup_iter=warmup)
            model.feed_data(data)
            model.optimize_parameters(current_iter)
            if current_iter % print_freq == 0:
                log = model.get_current_log()
                dt = (time.time() - t_iter) / print_freq
                t_iter = time.time()
                logger.info(f'[epoch {epoch:3d}, iter {current_iter:8,d}, lr {model.get_current_learning_rate()[0]:.3e}, '
                            f'{dt * 1e3:.1f} ms/iter] ' + ' '.join(f'{k}: {v:.4e}' for k, v in log.items()))
            if current_iter % save_freq == 0:
                logger.info('Saving models and training states.')
                model.save(epoch, current_iter)
            if val_freq and val_loaders and current_iter % val_freq == 0:  # reference train.py:195-198
                for vl in val_loaders:
                    model.validation(vl, current_iter, None, (opt.get('val') or {}).get('save_img', False))
        if current_iter > total_iters:
            break
    logger.info('End of training. Save the latest model.')
    model.save(epoch=-1, current_iter=-1)
    if (opt.get('val') or {}).get('metrics') is not None:
        for vl in val_loaders:
            model.validation(vl, current_iter, None, (opt.get('val') or {}).get('save_img', False))
    return model


if __name__ == '__main__':
    train_pipeline(os.path.abspath(os.path.join(os.path.dirname(__file__), os.pardir)))

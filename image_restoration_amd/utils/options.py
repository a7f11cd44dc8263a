"""YAML option files -> nested OrderedDict with the derived fields the models expect.

Counterpart of basicsr/utils/options.py (parse :37-95, dict2str :98-116, parse_options :119-150): same keys
(`name, model_type, scale, num_gpu, manual_seed, datasets, network_g, network_d, path, train, val, logger,
dist_params`), same derived paths (experiments/<name>/{models,training_states,visualization}, results/<name>),
same CLI (-opt, --launcher {none,pytorch}, --auto_resume, --debug, --local_rank), debug-mode interval overrides,
seed + rank seeding.
"""
import argparse
import os
import random
from collections import OrderedDict

import numpy as np
import torch
import yaml

from .dist_util import get_dist_info, init_dist


def _ordered(node):
    if isinstance(node, dict):
        return OrderedDict((k, _ordered(v)) for k, v in node.items())
    if isinstance(node, list):
        return [_ordered(v) for v in node]
    return node


def load_yaml(path):
    with open(path, 'r') as f:
        return _ordered(yaml.safe_load(f))


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def parse(opt_path, root_path, is_train=True, debug=False):
    opt = load_yaml(opt_path)
    if debug and not opt['name'].startswith('debug'):
        opt['name'] = 'debug_' + opt['name']
    opt['is_train'] = is_train
    if opt['num_gpu'] == 'auto':
        opt['num_gpu'] = torch.cuda.device_count()
    for phase, dataset in (opt.get('datasets') or {}).items():
        dataset['phase'] = phase.split('_')[0]  # test_1, test_2 -> test
        if 'scale' in opt:
            dataset['scale'] = opt['scale']
        for key in ('dataroot_gt', 'dataroot_lq'):
            if dataset.get(key) is not None:
                dataset[key] = os.path.expanduser(dataset[key])
    opt.setdefault('path', OrderedDict())
    for key, val in opt['path'].items():
        if val is not None and ('resume_state' in key or 'pretrain_network' in key):
            opt['path'][key] = os.path.expanduser(val)
    if is_train:
        root = os.path.join(root_path, 'experiments', opt['name'])
        opt['path'].update(experiments_root=root, models=os.path.join(root, 'models'),
                           training_states=os.path.join(root, 'training_states'), log=root,
                           visualization=os.path.join(root, 'visualization'))
        if 'debug' in opt['name']:
            if 'val' in opt:
                opt['val']['val_freq'] = 8
            opt['logger']['print_freq'] = 1
            opt['logger']['save_checkpoint_freq'] = 8
    else:
        root = os.path.join(root_path, 'results', opt['name'])
        opt['path'].update(results_root=root, log=root, visualization=os.path.join(root, 'visualization'))
    return opt


def dict2str(opt, indent_level=1):
    msg = '\n'
    pad = ' ' * (indent_level * 2)
    for k, v in opt.items():
        if isinstance(v, dict):
            msg += f'{pad}{k}:[{dict2str(v, indent_level + 1)}{pad}]\n'
        else:
            msg += f'{pad}{k}: {v}\n'
    return msg


def parse_options(root_path, is_train=True, argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument('-opt', type=str, required=True, help='Path to option YAML file.')
    parser.add_argument('--launcher', choices=['none', 'pytorch'], default='none', help='job launcher')
    parser.add_argument('--auto_resume', action='store_true')
    parser.add_argument('--debug', action='store_true')
    parser.add_argument('--local_rank', type=int, default=0)
    args = parser.parse_args(argv)
    opt = parse(args.opt, root_path, is_train=is_train, debug=args.debug)
    opt['auto_resume'] = args.auto_resume
    if args.launcher == 'none':
        opt['dist'] = False
    else:
        opt['dist'] = True
        # like the reference, only the backend of dist_params matters for the pytorch launcher (options.py:136-139)
        init_dist(args.launcher, backend=(opt.get('dist_params') or {}).get('backend', 'nccl'))
    opt['rank'], opt['world_size'] = get_dist_info()
    seed = opt.get('manual_seed')
    if seed is None:
        seed = random.randint(1, 10000)
        opt['manual_seed'] = seed
    set_random_seed(seed + opt['rank'])
    return opt

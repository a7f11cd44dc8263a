"""YAML option files -> nested OrderedDict plus the derived fields the models expect.

Behaviour of basicsr/utils/options.py (parse :37-95, dict2str :98-116, parse_options :119-150), pinned against the reference's own
output on this repository's option files (golden G-r): same keys (``name, model_type, scale, num_gpu, manual_seed, datasets,
network_g, network_d, path, train, val, logger, dist_params``), same derived run directories (experiments/<name>/{models,
training_states,visualization}, results/<name>), same debug-mode overrides, same command line (-opt, --launcher {none,pytorch},
--auto_resume, --debug, --local_rank) and ``manual_seed + rank`` seeding.  ``parse`` is a sequence of small passes over the
loaded tree instead of one long function.
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
    """yaml.safe_load keeps file order in plain dicts; the models expect OrderedDicts all the way down."""
    if isinstance(node, dict):
        return OrderedDict((key, _ordered(value)) for key, value in node.items())
    if isinstance(node, list):
        return [_ordered(value) for value in node]
    return node


def load_yaml(path):
    with open(path) as f:
        return _ordered(yaml.safe_load(f))


def set_random_seed(seed):
    """Python, numpy and torch (host and every visible device) generators."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


# ---------------------------------------------------------------------------------------------- passes of parse()
def _pass_datasets(opt):
    """Each ``datasets.<phase>[_k]`` block learns its phase (``test_1`` -> ``test``) and the global scale; ``~`` is expanded in
    its data roots."""
    for block_name, block in (opt.get('datasets') or {}).items():
        block['phase'] = block_name.split('_')[0]
        if 'scale' in opt:
            block['scale'] = opt['scale']
        for root in ('dataroot_gt', 'dataroot_lq'):
            if block.get(root) is not None:
                block[root] = os.path.expanduser(block[root])


def _pass_user_paths(opt):
    opt.setdefault('path', OrderedDict())
    paths = opt['path']
    for key in list(paths):
        if paths[key] is not None and ('resume_state' in key or 'pretrain_network' in key):
            paths[key] = os.path.expanduser(paths[key])


def _pass_run_directories(opt, root_path, is_train):
    paths = opt['path']
    if not is_train:
        base = os.path.join(root_path, 'results', opt['name'])
        paths['results_root'] = base
        paths['log'] = base
        paths['visualization'] = os.path.join(base, 'visualization')
        return
    base = os.path.join(root_path, 'experiments', opt['name'])
    paths['experiments_root'] = base
    for leaf in ('models', 'training_states'):
        paths[leaf] = os.path.join(base, leaf)
    paths['log'] = base
    paths['visualization'] = os.path.join(base, 'visualization')
    if 'debug' in opt['name']:       # short intervals so a debug run exercises validation, logging and checkpoints quickly
        if 'val' in opt:
            opt['val']['val_freq'] = 8
        opt['logger']['print_freq'] = 1
        opt['logger']['save_checkpoint_freq'] = 8


def parse(opt_path, root_path, is_train=True, debug=False):
    opt = load_yaml(opt_path)
    if debug and not opt['name'].startswith('debug'):
        opt['name'] = 'debug_' + opt['name']
    opt['is_train'] = is_train
    if opt['num_gpu'] == 'auto':
        opt['num_gpu'] = torch.cuda.device_count()
    _pass_datasets(opt)
    _pass_user_paths(opt)
    _pass_run_directories(opt, root_path, is_train)
    return opt


def dict2str(opt, indent_level=1):
    """The reference's printed form of an option tree: ``key: value`` lines, nested blocks in ``key:[ ... ]``, two spaces per level."""
    pad = '  ' * indent_level
    rows = ['\n']
    for key, value in opt.items():
        if isinstance(value, dict):
            rows.append(f'{pad}{key}:[{dict2str(value, indent_level + 1)}{pad}]\n')
        else:
            rows.append(f'{pad}{key}: {value}\n')
    return ''.join(rows)


def _command_line(argv):
    cli = argparse.ArgumentParser()
    cli.add_argument('-opt', type=str, required=True, help='Path to option YAML file.')
    cli.add_argument('--launcher', choices=['none', 'pytorch'], default='none', help='job launcher')
    cli.add_argument('--auto_resume', action='store_true')
    cli.add_argument('--debug', action='store_true')
    cli.add_argument('--local_rank', type=int, default=0)
    return cli.parse_args(argv)


def parse_options(root_path, is_train=True, argv=None):
    args = _command_line(argv)
    opt = parse(args.opt, root_path, is_train=is_train, debug=args.debug)
    opt['auto_resume'] = args.auto_resume
    opt['dist'] = args.launcher != 'none'
    if opt['dist']:
        # like the reference, only the backend of dist_params matters for the pytorch launcher (options.py:136-139)
        init_dist(args.launcher, backend=(opt.get('dist_params') or {}).get('backend', 'nccl'))
    opt['rank'], opt['world_size'] = get_dist_info()
    if opt.get('manual_seed') is None:
        opt['manual_seed'] = random.randint(1, 10000)
    # per-rank streams for data order / augmentation; the NETWORKS are made equal by BaseModel.align_replicas afterwards
    set_random_seed(opt['manual_seed'] + opt['rank'])
    return opt

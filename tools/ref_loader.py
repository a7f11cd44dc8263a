"""Synthetic-package loader for the read-only reference (build container only).

The reference package's ``__init__`` star-imports cv2/torchvision-dependent
modules that are absent here, so the leaf modules on the RRDBNet/ESRGAN path
are imported through empty parent packages whose ``__path__`` points into
``/root/reference``.  Nothing from the reference is copied; this
file only arranges ``sys.modules`` so the reference's own files can be executed
in place to produce golden vectors.  It never runs on the GPU box (the
reference does not travel): tests read the committed fixtures instead.
"""
import importlib
import logging
import os
import sys
import types
from copy import deepcopy

REF_ROOT = os.environ.get('SR_REFERENCE_ROOT', '/root/reference')
CPR = os.path.join(REF_ROOT, 'Car_Plate-Restoration')


def _pkg(name, path=None):
    m = types.ModuleType(name)
    m.__path__ = [path] if path else []
    m.__package__ = name
    sys.modules[name] = m
    return m


def load_reference():
    """Returns a namespace with the reference's path classes/registries."""
    if not os.path.isdir(CPR):
        raise FileNotFoundError(f'reference not mounted at {CPR}')
    sys.dont_write_bytecode = True
    if 'basicsr' in sys.modules and getattr(sys.modules['basicsr'], '_sr_synthetic', False):
        return sys.modules['basicsr']._sr_ns
    b = _pkg('basicsr', os.path.join(CPR, 'basicsr'))
    b._sr_synthetic = True
    utils = _pkg('basicsr.utils', os.path.join(CPR, 'basicsr', 'utils'))
    archs = _pkg('basicsr.archs', os.path.join(CPR, 'basicsr', 'archs'))
    _pkg('basicsr.ops', os.path.join(CPR, 'basicsr', 'ops'))
    losses = _pkg('basicsr.losses', os.path.join(CPR, 'basicsr', 'losses'))
    models = _pkg('basicsr.models', os.path.join(CPR, 'basicsr', 'models'))
    metrics = _pkg('basicsr.metrics')
    metrics.calculate_metric = None
    vgg = types.ModuleType('basicsr.archs.vgg_arch')
    vgg.VGGFeatureExtractor = object
    sys.modules['basicsr.archs.vgg_arch'] = vgg

    registry = importlib.import_module('basicsr.utils.registry')
    dist_util = importlib.import_module('basicsr.utils.dist_util')
    logger_mod = importlib.import_module('basicsr.utils.logger')
    utils.get_root_logger = logger_mod.get_root_logger
    utils.imwrite = None
    utils.tensor2img = None
    utils.master_only = dist_util.master_only
    logging.getLogger('basicsr').setLevel(logging.ERROR)

    rrdb = importlib.import_module('basicsr.archs.rrdbnet_arch')
    disc = importlib.import_module('basicsr.archs.discriminator_arch')
    arch_util = importlib.import_module('basicsr.archs.arch_util')

    def build_network(opt):
        opt = deepcopy(opt)
        return registry.ARCH_REGISTRY.get(opt.pop('type'))(**opt)

    archs.build_network = build_network
    loss_mod = importlib.import_module('basicsr.losses.losses')
    loss_util = importlib.import_module('basicsr.losses.loss_util')

    def build_loss(opt):
        opt = deepcopy(opt)
        return registry.LOSS_REGISTRY.get(opt.pop('type'))(**opt)

    losses.build_loss = build_loss
    lr_sched = importlib.import_module('basicsr.models.lr_scheduler')
    models.lr_scheduler = lr_sched
    sr_model = importlib.import_module('basicsr.models.sr_model')
    srgan_model = importlib.import_module('basicsr.models.srgan_model')
    esrgan_model = importlib.import_module('basicsr.models.esrgan_model')
    sampler = None
    try:
        _pkg('basicsr.data', os.path.join(CPR, 'basicsr', 'data'))
        sampler = importlib.import_module('basicsr.data.data_sampler')
    except Exception:  # pragma: no cover - optional
        sampler = None

    transforms = None
    try:  # paired_random_crop / mod_crop are plain numpy; the module's `import cv2` needs a name to bind (cv2 is not installed):
        # an EMPTY placeholder module — the functions that call into cv2 (augment: cv2.flip) are never used through it
        placeholder = 'cv2' not in sys.modules
        if placeholder:
            sys.modules['cv2'] = types.ModuleType('cv2')
        try:
            transforms = importlib.import_module('basicsr.data.transforms')
        finally:
            if placeholder:
                del sys.modules['cv2']
    except Exception:  # pragma: no cover - optional
        transforms = None

    data_util = None
    try:  # paired_paths_from_folder / _from_meta_info_file are plain Python over scandir (misc.py); same cv2 placeholder
        misc = importlib.import_module('basicsr.utils.misc')
        utils.scandir = misc.scandir
        utils.img2tensor = None
        placeholder = 'cv2' not in sys.modules
        if placeholder:
            sys.modules['cv2'] = types.ModuleType('cv2')
        try:
            data_util = importlib.import_module('basicsr.data.data_util')
        finally:
            if placeholder:
                del sys.modules['cv2']
    except Exception:  # pragma: no cover - optional
        data_util = None

    options_mod = None
    try:  # options.parse / dict2str: needs `set_random_seed` by name only
        utils.set_random_seed = lambda seed: None
        options_mod = importlib.import_module('basicsr.utils.options')
    except Exception:  # pragma: no cover - optional
        options_mod = None

    psnr_mod = None
    try:  # calculate_psnr and the y-channel conversion are plain numpy (same empty cv2 placeholder as above; calculate_ssim,
        # which does call cv2, is not used)
        metrics.__path__ = [os.path.join(CPR, 'basicsr', 'metrics')]
        placeholder = 'cv2' not in sys.modules
        if placeholder:
            sys.modules['cv2'] = types.ModuleType('cv2')
        try:
            psnr_mod = importlib.import_module('basicsr.metrics.psnr_ssim')
        finally:
            if placeholder:
                del sys.modules['cv2']
    except Exception:  # pragma: no cover - optional
        psnr_mod = None

    ns = types.SimpleNamespace(
        registry=registry, RRDBNet=rrdb.RRDBNet, RRDB=rrdb.RRDB, ResidualDenseBlock=rrdb.ResidualDenseBlock,
        VGGStyleDiscriminator128=disc.VGGStyleDiscriminator128, VGGStyleDiscriminator256=disc.VGGStyleDiscriminator256,
        pixel_unshuffle=arch_util.pixel_unshuffle,
        L1Loss=loss_mod.L1Loss, GANLoss=loss_mod.GANLoss, l1_loss=loss_mod.l1_loss, loss_util=loss_util,
        MSELoss=loss_mod.MSELoss, CharbonnierLoss=loss_mod.CharbonnierLoss,
        lr_scheduler=lr_sched, SRModel=sr_model.SRModel, SRGANModel=srgan_model.SRGANModel,
        ESRGANModel=esrgan_model.ESRGANModel, data_sampler=sampler, build_network=build_network, transforms=transforms,
        calculate_psnr=getattr(psnr_mod, 'calculate_psnr', None), options=options_mod, data_util=data_util)
    b._sr_ns = ns
    return ns

"""Drop-in boundary, CPU side: registry semantics, option surface, state_dict contract, and
that libsr_hip.so loads and exports every symbol include/sr_hip.h declares (no compute)."""
import ctypes
import os
import re

import pytest
import torch

import image_restoration_amd as ira
from image_restoration_amd import _lib
from image_restoration_amd.utils import synth
from image_restoration_amd.utils.registry import ARCH_REGISTRY, Registry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_registry_semantics():
    r = Registry('t')

    @r.register()
    class A:
        pass

    class B:
        pass

    r.register(B)
    assert r.get('A') is A and r.get('B') is B and 'A' in r and set(r.keys()) == {'A', 'B'}
    with pytest.raises(KeyError):
        r.get('missing')
    with pytest.raises(AssertionError):
        r.register(A)


def test_build_network_and_state_dict_contract():
    opt = dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, num_feat=64, num_block=23, num_grow_ch=32)
    net = ira.build_network(opt)
    assert opt['type'] == 'RRDBNet'  # deepcopy: caller's dict untouched
    assert 'RRDBNet' in ARCH_REGISTRY
    sd = net.state_dict()
    assert len(sd) == 702 and sum(p.numel() for p in net.parameters()) == 16697987
    expected = [n for n, _ in synth.rrdbnet_param_shapes(3, 3, 4, 64, 23, 32)]
    assert list(sd.keys()) == expected
    for (n, shape) in synth.rrdbnet_param_shapes(3, 3, 4, 64, 23, 32):
        assert tuple(sd[n].shape) == shape
    # checkpoints with the reference's keys load strictly
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(0, num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32).items()}, strict=True)
    with pytest.raises(KeyError):
        ira.build_network(dict(type='NoSuchArch'))


def test_rdb_init_matches_reference_statistics():
    # RDB convs: kaiming_normal * 0.1, bias 0 (arch_util.py:12-40 via rrdbnet_arch.py:30); other convs keep
    # the nn.Conv2d default (uniform, |w| <= 1/sqrt(fan_in)).
    torch.manual_seed(0)
    net = ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, num_feat=64, num_block=1))
    w = net.body[0].rdb1.conv5.weight
    fan_in = 192 * 9
    assert abs(float(w.std()) - 0.1 * (2.0 / fan_in) ** 0.5) / (0.1 * (2.0 / fan_in) ** 0.5) < 0.05
    assert float(net.body[0].rdb1.conv5.bias.abs().max()) == 0.0
    assert float(net.conv_body.weight.abs().max()) <= 1 / (64 * 9) ** 0.5 + 1e-7
    assert float(net.conv_body.bias.abs().max()) > 0


def test_scale_changes_first_conv_width():
    assert ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=2, num_feat=16, num_block=1,
                                  num_grow_ch=8)).conv_first.weight.shape[1] == 12
    assert ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=1, num_feat=16, num_block=1,
                                  num_grow_ch=8)).conv_first.weight.shape[1] == 48


def test_no_cpu_fallback():
    net = ira.build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, num_feat=16, num_block=1, num_grow_ch=8))
    with pytest.raises(_lib.SrHipError):
        net(torch.rand(1, 3, 8, 8))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, 'include', 'sr_hip.h')).read()
    declared = set(re.findall(r'\b(sr_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sr_version() == _lib.SR_ABI_VERSION
    # host-only helpers are callable without a GPU
    cfg = _lib.RRDBNetCfg(3, 3, 4, 64, 23, 32)
    assert lib.sr_rrdbnet_num_params(ctypes.byref(cfg)) == 702
    assert lib.sr_rrdbnet_packed_bytes(ctypes.byref(cfg)) >= 16697987 * 4
    assert lib.sr_rrdbnet_workspace_bytes(ctypes.byref(cfg), 1, 128, 128) > 0
    assert lib.sr_conv3x3_cin_pad(160, 64, 32) == 160 and lib.sr_conv3x3_cin_pad(44, 20, 12) == 24 + 2 * 16
    assert lib.sr_conv3x3_cin_pad(45, 20, 12) < 0
    bad = _lib.RRDBNetCfg(3, 3, 3, 64, 23, 32)
    assert lib.sr_rrdbnet_num_params(ctypes.byref(bad)) < 0


def test_argument_errors_are_codes_with_messages_not_crashes():
    """Bad arguments come back as negative status codes with a message in sr_last_error() before anything is launched (so
    this runs without a GPU): null pointers, misaligned channel counts, bad configurations, out-of-range knobs."""
    lib = _lib.load()

    def failed(rc, word):
        msg = lib.sr_last_error().decode()
        return rc < 0 and word in msg
    assert failed(lib.sr_conv3x3_f32(None, None), 'null')
    d = _lib.ConvDesc()
    d.in_, d.wpacked, d.out = 256, 256, 256  # non-null, aligned placeholders: the shape checks come first
    d.cin_pad, d.cout, d.n, d.in_h, d.in_w = 12, 32, 1, 8, 8
    assert failed(lib.sr_conv3x3_f32(ctypes.byref(d), None), 'multiple of 8')
    d.cin_pad = 24
    assert failed(lib.sr_conv3x3_bf16(ctypes.byref(d), None), 'multiple of 16')
    d.cin_pad, d.s2_channels = 64, 32
    assert failed(lib.sr_conv3x3_bf16(ctypes.byref(d), None), 's2_channels')
    assert failed(lib.sr_cb16_add_bf16(256, 256, 256, 12, None), 'multiple of 8')
    assert failed(lib.sr_set_forward_groups(9), '1..4')
    assert lib.sr_set_forward_groups(0) == 0
    bad = _lib.RRDBNetCfg(3, 3, 3, 64, 23, 32)
    assert failed(lib.sr_rrdbnet_forward_f32(ctypes.byref(bad), 256, 256, 256, 1, 8, 8, 256, 1 << 20, None), 'config')
    good = _lib.RRDBNetCfg(3, 3, 2, 16, 1, 8)
    assert failed(lib.sr_rrdbnet_forward_f32(ctypes.byref(good), 256, 256, 256, 1, 9, 8, 256, 1 << 30, None), 'divisible')
    assert failed(lib.sr_gan_point_loss_fwd_f32(256, 16, 9, 0.0, 1.0, 256, 256, 1 << 20, None), 'bad argument')


def test_whole_discriminator_driver_plans_without_a_gpu():
    """sr_vgg_*: the host-only helpers (plan, sizes) and the argument checks of the drivers — no launch happens."""
    lib = _lib.load()
    c128, c256 = _lib.VGGCfg(3, 64, 128), _lib.VGGCfg(3, 64, 256)
    assert lib.sr_vgg_num_params(ctypes.byref(c128)) == 33 and lib.sr_vgg_num_batchnorm(ctypes.byref(c128)) == 9
    assert lib.sr_vgg_num_params(ctypes.byref(c256)) == 39 and lib.sr_vgg_num_batchnorm(ctypes.byref(c256)) == 11
    # the state_dict of the module has exactly those tensors, in that order of kinds
    net = ira.build_network(dict(type='VGGStyleDiscriminator128', num_in_ch=3, num_feat=64))
    assert len(list(net.parameters())) == 33 and len(list(net.buffers())) == 27
    assert [k for k, _ in net.named_parameters()][:5] == ['conv0_0.weight', 'conv0_0.bias', 'conv0_1.weight', 'bn0_1.weight', 'bn0_1.bias']
    assert [k for k, _ in net.named_parameters()][-4:] == ['linear1.weight', 'linear1.bias', 'linear2.weight', 'linear2.bias']
    for suffix in ('', '_bf16'):
        packed = getattr(lib, 'sr_vgg_packed_bytes' + suffix)(ctypes.byref(c128))
        s1, s2 = (getattr(lib, 'sr_vgg_saved_bytes' + suffix)(ctypes.byref(c128), n) for n in (1, 2))
        w1 = getattr(lib, 'sr_vgg_workspace_bytes' + suffix)(ctypes.byref(c128), 1)
        assert packed > 0 and 0 < s1 < s2 < 2 * s1 + (1 << 20) and w1 > 0
    # fp32 weights of the 128 network: every conv twice (forward + data-gradient image) is at least 2 x the raw conv weights
    raw = sum(p.numel() for k, p in net.named_parameters() if k.startswith('conv')) * 4
    assert lib.sr_vgg_packed_bytes(ctypes.byref(c128)) >= 2 * raw
    assert lib.sr_vgg_num_params(ctypes.byref(_lib.VGGCfg(3, 64, 96))) == 0              # unsupported input size
    assert lib.sr_vgg_packed_bytes_bf16(ctypes.byref(_lib.VGGCfg(3, 8, 128))) == 0          # bf16 needs num_feat % 16 == 0
    assert lib.sr_vgg_saved_bytes(ctypes.byref(c128), 0) == 0

    def failed(rc, word):
        return rc < 0 and word in lib.sr_last_error().decode()
    assert failed(lib.sr_vgg_forward_f32(ctypes.byref(c128), None, None, None, None, None, 1, 1, None, 0, None, 0, None), 'null')
    bad = _lib.VGGCfg(3, 64, 100)
    assert failed(lib.sr_vgg_backward_f32(ctypes.byref(bad), None, None, None, 0, None, 1, 1, None, 0, None, None, 0, None), 'bad configuration')
    assert failed(lib.sr_adam_step_f32(None, None, None, None, 0, 1, 1e-4, 0.9, 0.99, 1e-8, 0.0, 1.0, None, None, None), 'bad argument')

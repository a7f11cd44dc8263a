"""The oracle (oracle/rrdbnet_ref.py) against the golden vectors produced by the reference
itself (tools/make_goldens.py).  CPU only.  Tolerances: the oracle runs the same ATen CPU
kernels as the reference, so agreement is to rounding (1e-6 abs on O(1) values)."""
import numpy as np
import torch

from image_restoration_amd.utils import synth
from oracle import rrdbnet_ref as R

TOL = 2e-6


def _close(a, b, tol=TOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b).max()
    assert err <= tol, f'max abs err {err}'


def test_rdb_forward_and_intermediates(golden):
    g = golden('g_a_rdb')
    sd = synth.rdb_state_dict(11, 64, 32)
    out, (x1, x2, x3, x4) = R.rdb_forward(torch.from_numpy(g['x']), sd, return_intermediates=True)
    for got, key in ((out, 'out'), (x1, 'x1'), (x2, 'x2'), (x3, 'x3'), (x4, 'x4')):
        _close(got, g[key])


def test_rdb_backward(golden):
    g = golden('g_a_rdb')
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.rdb_state_dict(11, 64, 32).items()}
    x = torch.from_numpy(g['x']).requires_grad_(True)
    (R.rdb_forward(x, sd) * torch.from_numpy(g['R'])).sum().backward()
    _close(x.grad, g['grad_x'], 1e-5)
    for k, v in sd.items():
        _close(v.grad, g['grad_' + k.replace('.', '_')], 2e-5)


def test_rrdb(golden):
    g = golden('g_c_rrdb')
    sd = synth.rrdb_state_dict(21, 64, 32)
    _close(R.rrdb_forward(torch.from_numpy(g['x']), sd), g['out'])


def test_config1_full_network(golden):
    g = golden('g_d_c1')
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=32, num_block=1, num_grow_ch=32)
    sd = synth.rrdbnet_state_dict(0, **cfg)
    _close(R.rrdbnet_forward(g['x'], sd, 4, 1), g['y'], 5e-6)


def test_full23(golden):
    g = golden('g_e_full23')
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32)
    sd = synth.rrdbnet_state_dict(0, **cfg)
    _close(R.rrdbnet_forward(g['x'], sd, 4, 23), g['y'], 1e-5)


def test_head_and_ragged(golden):
    g = golden('g_f_head')
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=0, num_grow_ch=32)
    sd = synth.rrdbnet_state_dict(3, **cfg)
    _close(R.head_forward(torch.from_numpy(g['feat']), sd), g['out'], 5e-6)
    _close(R.rrdbnet_forward(g['x'], sd, 4, 0), g['y'], 5e-6)


def test_scale_2_and_1(golden):
    g = golden('g_l_scale')
    for scale in (2, 1):
        cfg = dict(num_in_ch=3, num_out_ch=3, scale=scale, num_feat=16, num_block=1, num_grow_ch=8)
        sd = synth.rrdbnet_state_dict(40 + scale, **cfg)
        _close(R.rrdbnet_forward(g[f'x_s{scale}'], sd, scale, 1), g[f'y_s{scale}'], 5e-6)
        u = 2 if scale == 2 else 4
        pu = R.pixel_unshuffle(torch.arange(2 * 3 * 8 * 8, dtype=torch.float32).view(2, 3, 8, 8), u)
        assert np.array_equal(pu.numpy(), g[f'unshuffle_s{scale}'])

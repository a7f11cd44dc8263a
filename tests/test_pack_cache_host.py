"""hip_ops.cached_pack: one weight image per (parameter, kind) until the parameter changes (no GPU needed: the builder is a stub)."""
import gc

import torch

from image_restoration_amd import hip_ops as H


def test_cached_pack_follows_parameter_identity_version_and_epochs():
    w = torch.nn.Parameter(torch.ones(2, 2, 3, 3))
    b = torch.nn.Parameter(torch.zeros(2))
    built = []

    def build():
        built.append(1)
        return object()
    first = H.cached_pack('fwd', w, b, build)
    assert H.cached_pack('fwd', w, b, build) is first and len(built) == 1
    assert H.cached_pack('dgrad', w, None, build) is not first and len(built) == 2       # kinds are separate entries
    with torch.no_grad():
        w.add_(1)                                                                        # torch's version counter
    assert H.cached_pack('fwd', w, b, build) is not first and len(built) == 3
    with torch.no_grad():
        b.add_(1)                                                                        # ... the bias's too
    H.cached_pack('fwd', w, b, build)
    assert len(built) == 4
    cell = [0]
    w._sr_epoch = cell                                                                   # an optimiser that writes through raw pointers
    H.cached_pack('fwd', w, b, build)
    n = len(built)
    cell[0] += 1
    H.cached_pack('fwd', w, b, build)
    assert len(built) == n + 1
    H.invalidate_packs()                                                                 # the global form
    H.cached_pack('fwd', w, b, build)
    assert len(built) == n + 2
    H.cached_pack('fwd', w, b, build)
    assert len(built) == n + 2


def test_cached_pack_ignores_temporaries_and_forgets_dead_parameters():
    built = []

    def build():
        built.append(1)
        return object()
    t = torch.ones(2, 2, 3, 3)                    # not a Parameter (a spectrally normalised weight): built every time, nothing kept
    n0 = len(H._pack_cache)
    H.cached_pack('fwd', t, None, build)
    H.cached_pack('fwd', t, None, build)
    assert len(built) == 2 and len(H._pack_cache) == n0
    w = torch.nn.Parameter(torch.ones(2, 2, 3, 3))
    H.cached_pack('fwd', w, None, build)
    assert len(H._pack_cache) == n0 + 1
    del w
    gc.collect()
    assert len(H._pack_cache) == n0               # the entry (and the images it held) went with the parameter

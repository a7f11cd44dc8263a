"""sr_conv3x3_chain_bf16: the five convs of a residual dense block (rrdbnet_arch.py:32-39) as ONE persistent launch with tile-level
hand-offs, against the same convs launched one by one (sr_conv3x3_bf16).  Both run the same tile code on the same operands in the
same order, so the results must be BIT-identical; what the test probes is the hand-off (write-through stores, agent-scope flags,
acquire before the dependent loads) — a stale or early read shows up as a difference.  Shapes cover the one-launch path (>= 256
tiles of 32x32), batches with more tiles than resident workgroups, ragged widths, repeated calls on one sync block (increasing
call_index) and the fallbacks (small launches, ragged heights), and a run under uneven load from a second stream.
Mode 3 is the fused dense-block kernel (rdb_fused_bf16_kernel: one workgroup per tile for the whole block, partial sums of all
unfinished convs in registers, neighbour hand-offs at every conv); shapes it does not take fall back to mode 2 inside the entry point."""
import pytest
import torch

from image_restoration_amd import _lib
from image_restoration_amd import hip_ops as H

pytestmark = pytest.mark.gpu


WAVE4_DEFAULT = 0   # the library's default for sr_dev_set_fused_wave4 (csrc/conv_bf16.hip: g_fused_wave4)


@pytest.fixture(autouse=True, params=['waves8', 'waves4'])
def chain_restore(request):
    """Both tile geometries of the chain launch (sr_set_conv_chain 1: 32-row ring tiles, 2: 16-row tiles) and the fused kernel (3, the
    library's default, restored afterwards so that the tests that follow in the same process run what the product runs).  Every test
    that reaches the fused kernel runs twice: its 16-row tiles on eight waves of two rows and on four waves of four rows
    (rdb_fused4_bf16_kernel, `sr_dev_set_fused_wave4`)."""
    import ctypes as C
    lib = _lib.load()
    lib.sr_dev_set_fused_wave4.argtypes = [C.c_int]
    lib.sr_dev_set_fused_wave4.restype = None
    w4 = request.param == 'waves4'
    callspec = getattr(request.node, 'callspec', None)
    if w4 and callspec is not None and callspec.params.get('mode', 3) != 3:
        pytest.skip('the four-wave instance is a variant of the fused kernel (mode 3) only')
    lib.sr_dev_set_fused_wave4(1 if w4 else 0)
    yield
    lib.sr_dev_set_fused_wave4(WAVE4_DEFAULT)
    _lib.check(lib.sr_set_conv_chain(3), 'sr_set_conv_chain')


def _rdb(dev, nf, gc, seed):
    g = torch.Generator().manual_seed(seed)
    packs = []
    for k in range(1, 6):
        cout, cin = (nf if k == 5 else gc), nf + (k - 1) * gc
        w = (torch.randn(cout, cin, 3, 3, generator=g) * (0.6 / (cin * 9) ** 0.5)).to(dev)
        b = (torch.randn(cout, generator=g) * 0.05).to(dev)
        packs.append(H.PackedConvBF16(w, b, first_seg=nf, seg=gc))
    return packs


def _steps(cat, nxt, packs, nf, gc, last_res2=None):
    steps = []
    for k in range(1, 5):
        steps.append((cat.slice(0, nf + (k - 1) * gc), packs[k - 1], cat.slice(nf + (k - 1) * gc, gc), dict(act_slope=0.2)))
    kw = dict(alpha=0.2, res1=cat.slice(0, nf), beta1=1.0)
    if last_res2 is not None:
        kw = dict(alpha=0.04, res1=cat.slice(0, nf), beta1=0.2, res2=last_res2, beta2=1.0)
    steps.append((cat, packs[4], nxt.slice(0, nf), kw))
    return steps


def _fresh(dev, n, nf, gc, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    ctot = nf + 4 * gc
    buf = torch.zeros(n, ctot // 16, h, w, 16, dtype=torch.bfloat16)
    buf[:, :nf // 16] = torch.randn(n, nf // 16, h, w, 16, generator=g).to(torch.bfloat16)
    buf[:, nf // 16:] = 7.0   # poison: a conv that reads x_k before it was produced sees this, not zeros
    return H.CB16(buf.to(dev)), H.CB16(torch.full((n, ctot // 16, h, w, 16), -3.0, dtype=torch.bfloat16, device=dev))


@pytest.mark.parametrize('n,h,w,nf,gc', [
    (16, 128, 128, 64, 32),    # BASELINE config 2's dense block: 256 tiles of 32x32 = one per CU
    (40, 128, 96, 64, 32),     # 480 tiles: every workgroup claims several items per conv
    (12, 160, 100, 64, 32),    # ragged width (last tile column 4 px wide), 240 tiles
    (64, 64, 64, 32, 32),      # nf = 32: all five convs on the 32-cout tile
    (2, 64, 40, 64, 32),       # small launch: conv-by-conv fallback inside the entry point
    (12, 120, 128, 64, 32),    # height not a multiple of 32: fallback
    (20, 128, 128, 64, 32),    # fused kernel: three rounds of 8 images, the last one partly empty
    (5, 208, 96, 64, 32),      # fused kernel: 39 tiles per image, a window of 195 tiles
    (2, 544, 544, 64, 32),     # fused kernel: a tiler cell = 578 tiles per image, more than CUs: the window slides down the image
    (3, 320, 1088, 64, 32),    # fused kernel: 34 tiles per row, 680 per image
    (32, 32, 32, 64, 32),      # the recipe's batch of 32x32 patches: 64 tiles
    (6, 48, 72, 64, 32),       # ragged width, 3 tile rows per image
])
@pytest.mark.parametrize('mode', [1, 2, 3])
def test_chain_equals_conv_by_conv_bit_for_bit(cuda, n, h, w, nf, gc, mode):
    _lib.check(_lib.load().sr_set_conv_chain(mode), 'sr_set_conv_chain')
    packs = _rdb(cuda, nf, gc, 3)
    cat_a, nxt_a = _fresh(cuda, n, nf, gc, h, w, 5)
    cat_b, nxt_b = _fresh(cuda, n, nf, gc, h, w, 5)
    for src, pc, out, kw in _steps(cat_a, nxt_a, packs, nf, gc):
        H.conv3x3_bf16(src, pc, out, **kw)
    sync = None
    for rep in range(3):       # the same sync block serves consecutive calls (a forward's dense blocks)
        if rep:
            cat_b, nxt_b = _fresh(cuda, n, nf, gc, h, w, 5)
        _, sync = H.conv3x3_chain_bf16(_steps(cat_b, nxt_b, packs, nf, gc), sync, call_index=rep)
        torch.cuda.synchronize()
        assert int(sync[0]) == 0, 'a dependency wait timed out'
        assert torch.equal(cat_a.buf, cat_b.buf), rep
        assert torch.equal(nxt_a.buf[:, :nf // 16], nxt_b.buf[:, :nf // 16]), rep
    assert bool(torch.isfinite(nxt_b.buf[:, :nf // 16].float()).all())


@pytest.mark.parametrize('n,h,w', [(32, 32, 32), (6, 48, 72), (2, 64, 40)])
def test_fused_block_on_8_row_tiles_equals_conv_by_conv_bit_for_bit(cuda, n, h, w):
    """The fused kernel's second instance (one row per wave: 8-row tiles, a chunk per step; `sr_dev_set_fused_rows8`: 1 = forward
    blocks of small launches, the default; 2 = transposed blocks too) against conv-by-conv launches, forward block."""
    import ctypes as C
    lib = _lib.load()
    lib.sr_dev_set_fused_rows8.argtypes = [C.c_int]
    lib.sr_dev_set_fused_rows8.restype = None
    nf, gc = 64, 32
    _lib.check(lib.sr_set_conv_chain(3), 'sr_set_conv_chain')
    packs = _rdb(cuda, nf, gc, 3)
    cat_a, nxt_a = _fresh(cuda, n, nf, gc, h, w, 5)
    for src, pc, out, kw in _steps(cat_a, nxt_a, packs, nf, gc):
        H.conv3x3_bf16(src, pc, out, **kw)
    lib.sr_dev_set_fused_rows8(1)
    try:
        sync = None
        for rep in range(3):
            cat_b, nxt_b = _fresh(cuda, n, nf, gc, h, w, 5)
            _, sync = H.conv3x3_chain_bf16(_steps(cat_b, nxt_b, packs, nf, gc), sync, call_index=rep)
            torch.cuda.synchronize()
            assert int(sync[0]) == 0, 'a dependency wait timed out'
            assert torch.equal(cat_a.buf, cat_b.buf), rep
            assert torch.equal(nxt_a.buf[:, :nf // 16], nxt_b.buf[:, :nf // 16]), rep
    finally:
        lib.sr_dev_set_fused_rows8(1)


@pytest.mark.parametrize('n,h,w', [(16, 128, 128), (20, 128, 128), (12, 160, 100), (5, 208, 96), (2, 544, 544), (32, 32, 32), (6, 48, 72)])
def test_fused_block_whose_intermediates_are_scratch_gives_the_same_output(cuda, n, h, w):
    """The inference forward tells the dense block that nobody reads x1..x4 afterwards (rrdbnet_bf16.hip: `mids_scratch`; here through
    the development switch).  A tile's inside then reaches the next conv through the LDS only and just its outermost ring — what the
    neighbouring tiles fetch — is stored: the block's OUTPUT must still equal the conv-by-conv launches bit for bit, on whole and ragged
    tiles, 16-row and 8-row instances, repeated calls on one sync block."""
    import ctypes as C
    lib = _lib.load()
    lib.sr_dev_set_chain_mids_scratch.argtypes = [C.c_int]
    lib.sr_dev_set_chain_mids_scratch.restype = None
    nf, gc = 64, 32
    _lib.check(lib.sr_set_conv_chain(3), 'sr_set_conv_chain')
    packs = _rdb(cuda, nf, gc, 3)
    cat_a, nxt_a = _fresh(cuda, n, nf, gc, h, w, 5)
    for src, pc, out, kw in _steps(cat_a, nxt_a, packs, nf, gc):
        H.conv3x3_bf16(src, pc, out, **kw)
    lib.sr_dev_set_chain_mids_scratch(1)
    try:
        sync = None
        for rep in range(3):
            cat_b, nxt_b = _fresh(cuda, n, nf, gc, h, w, 5)
            _, sync = H.conv3x3_chain_bf16(_steps(cat_b, nxt_b, packs, nf, gc), sync, call_index=rep)
            torch.cuda.synchronize()
            assert int(sync[0]) == 0, 'a dependency wait timed out'
            assert torch.equal(nxt_a.buf[:, :nf // 16], nxt_b.buf[:, :nf // 16]), rep
            assert torch.equal(cat_a.buf[:, :nf // 16], cat_b.buf[:, :nf // 16]), 'the block input is not touched'
        if h % 16 == 0 and n * (h // 16) * ((w + 31) // 32) >= 128:   # (where the 16-row fused kernel runs) the inside really stayed on chip
            inner = cat_b.buf[:, nf // 16:, 1:15, 1:31].float()
            assert bool((inner == 7.0).all()), 'expected the poison of _fresh() where nothing needs to be stored'
    finally:
        lib.sr_dev_set_chain_mids_scratch(0)


@pytest.mark.parametrize('mode', [1, 2, 3])
def test_chain_under_uneven_load_and_with_rrdb_residuals(cuda, mode):
    """Three chained dense blocks = one RRDB (the third closes with the RRDB residual, rrdbnet_arch.py:58-63) while a second stream
    keeps part of the chip busy with unrelated bandwidth-heavy work: hand-offs must hold when workgroups are delayed unevenly."""
    _lib.check(_lib.load().sr_set_conv_chain(mode), 'sr_set_conv_chain')
    n, h, w, nf, gc = 16, 128, 128, 64, 32
    packs = [_rdb(cuda, nf, gc, 10 + r) for r in range(3)]

    def run(chain):
        bufs = [_fresh(cuda, n, nf, gc, h, w, 21)[0] for _ in range(4)]
        sync = None
        for r in range(3):
            steps = _steps(bufs[r], bufs[r + 1], packs[r], nf, gc, last_res2=bufs[0].slice(0, nf) if r == 2 else None)
            if chain:
                _, sync = H.conv3x3_chain_bf16(steps, sync, call_index=r)
            else:
                for src, pc, out, kw in steps:
                    H.conv3x3_bf16(src, pc, out, **kw)
        return bufs, sync

    ref, _ = run(False)
    side = torch.cuda.Stream()
    noise = torch.empty(64 << 20, dtype=torch.float32, device=cuda)
    for trial in range(4):
        with torch.cuda.stream(side):
            for _ in range(6 + 5 * trial):
                noise.mul_(1.0001).add_(0.5)
        got, sync = run(True)
        torch.cuda.synchronize()
        assert int(sync[0]) == 0
        for a, b in zip(ref, got):
            assert torch.equal(a.buf[:, :nf // 16], b.buf[:, :nf // 16]), trial
        assert torch.equal(ref[2].buf, got[2].buf), trial


def test_chain_can_be_switched_off(cuda):
    lib = _lib.load()
    n, h, w, nf, gc = 16, 64, 64, 64, 32
    packs = _rdb(cuda, nf, gc, 1)
    cat_a, nxt_a = _fresh(cuda, n, nf, gc, h, w, 2)
    cat_b, nxt_b = _fresh(cuda, n, nf, gc, h, w, 2)
    _lib.check(lib.sr_set_conv_chain(0), 'sr_set_conv_chain')
    H.conv3x3_chain_bf16(_steps(cat_a, nxt_a, packs, nf, gc))
    _lib.check(lib.sr_set_conv_chain(1), 'sr_set_conv_chain')
    H.conv3x3_chain_bf16(_steps(cat_b, nxt_b, packs, nf, gc))
    torch.cuda.synchronize()
    assert torch.equal(cat_a.buf, cat_b.buf) and torch.equal(nxt_a.buf[:, :4], nxt_b.buf[:, :4])


@pytest.mark.parametrize('n,h,w', [(16, 128, 128), (20, 128, 128), (5, 208, 96), (2, 64, 40), (32, 32, 32), (6, 48, 72)])
@pytest.mark.parametrize('mode', [2, 3])
@pytest.mark.parametrize('closing', [False, True])
def test_transposed_block_chain_equals_conv_by_conv_bit_for_bit(cuda, n, h, w, mode, closing):
    """The data-gradient side of a dense block (sr_rrdbnet_backward_bf16): the same chain shape over D = [dY5 | dY4 .. dY1], no bias, the
    LeakyReLU-backward mask of the forward activation on conv1-4, the residual sum(s) on conv5 — in mode 3 the fused kernel's lean
    transposed-block epilogues (masks and residual sources fetched ahead of the epilogues)."""
    import ctypes as C
    lib = _lib.load()
    lib.sr_dev_set_fused_rows8.argtypes = [C.c_int]
    lib.sr_dev_set_fused_rows8.restype = None
    _lib.check(lib.sr_set_conv_chain(mode), 'sr_set_conv_chain')
    # small launches in mode 3: the transposed block also on the 8-row instance (switch 2; the default, 1, keeps it on 16-row tiles)
    lib.sr_dev_set_fused_rows8(2 if (mode == 3 and n * h * w <= 32 * 32 * 32) else 1)
    nf, gc = 64, 32
    g = torch.Generator().manual_seed(7)
    packs = []
    for k in range(1, 6):
        cout, cin = (nf if k == 5 else gc), nf + (k - 1) * gc
        wt = (torch.randn(cout, cin, 3, 3, generator=g) * (0.6 / (cin * 9) ** 0.5)).to(cuda)
        packs.append(H.PackedConvBF16(wt, None, first_seg=nf, seg=gc))
    fwd = H.CB16(torch.randn(n, (nf + 4 * gc) // 16, h, w, 16, generator=g).to(torch.bfloat16).to(cuda))   # saved activations
    rrdb = H.CB16(torch.randn(n, nf // 16, h, w, 16, generator=g).to(torch.bfloat16).to(cuda))

    def steps(cat, nxt):
        st = []
        for k in range(1, 5):
            st.append((cat.slice(0, nf + (k - 1) * gc), packs[k - 1], cat.slice(nf + (k - 1) * gc, gc),
                       dict(mask=fwd.slice(nf + (4 - k) * gc, gc), mask_slope=0.2)))
        kw = dict(res1=cat.slice(0, nf), beta1=0.2 if closing else 1.0)
        if closing:
            kw.update(res2=rrdb, beta2=1.0)
        st.append((cat, packs[4], nxt.slice(0, nf), kw))
        return st

    cat_a, nxt_a = _fresh(cuda, n, nf, gc, h, w, 5)
    for src, pc, out, kw in steps(cat_a, nxt_a):
        H.conv3x3_bf16(src, pc, out, **kw)
    sync = None
    for rep in range(2):
        cat_b, nxt_b = _fresh(cuda, n, nf, gc, h, w, 5)
        _, sync = H.conv3x3_chain_bf16(steps(cat_b, nxt_b), sync, call_index=rep)
        torch.cuda.synchronize()
        assert int(sync[0]) == 0
        assert torch.equal(cat_a.buf, cat_b.buf), rep
        assert torch.equal(nxt_a.buf[:, :nf // 16], nxt_b.buf[:, :nf // 16]), rep
    lib.sr_dev_set_fused_rows8(1)


def test_handoff_watchdog_reports_a_raised_abort_word_once(cuda):
    """A dense-block launch whose tiles waited in vain raises its sync block's abort word; the network drivers copy those words to
    pinned host memory behind their launches and sr_chain_watchdog / the next driver call turns a raised one into an error
    (here: the plumbing, with a word raised by hand — a real time-out needs a GPU that withholds CUs)."""
    import ctypes as C
    lib = _lib.load()
    lib.sr_dev_chain_watch.argtypes = [C.c_void_p, C.c_void_p]
    lib.sr_dev_chain_watch.restype = None
    assert lib.sr_chain_watchdog() == 0
    word = torch.zeros(4, dtype=torch.int32, device=cuda)
    lib.sr_dev_chain_watch(word.data_ptr(), None)
    torch.cuda.synchronize()
    assert lib.sr_chain_watchdog() == 0
    word[0] = 1
    torch.cuda.synchronize()
    lib.sr_dev_chain_watch(word.data_ptr(), None)
    torch.cuda.synchronize()
    assert lib.sr_chain_watchdog() < 0 and b'timed out' in lib.sr_last_error()
    assert lib.sr_chain_watchdog() == 0

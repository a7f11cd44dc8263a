"""conv_stream_bf16_kernel — the streaming form of sr_conv3x3_bf16 for few input channels on a large pixel grid (resident weights, one
LDS-DMA pipeline across a strip of tiles, counted waits, the extra epilogue operand fetched a tile ahead) — against the per-tile
kernel on the same operands.  Both accumulate chunk by chunk, tap column by tap column, tap row by tap row and run the same epilogue
arithmetic, so the results must be BIT-identical; what the test probes is the pipeline: the ring running across tile and image
boundaries, strips of uneven length, the static accounting of what may be in flight at each wait."""
import pytest
import torch

from image_restoration_amd import _lib
from image_restoration_amd import hip_ops as H

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def stream_restore():
    yield
    _lib.load().sr_dev_set_conv_stream(1)


@pytest.mark.parametrize('cin,n,h,w,extra', [
    (64, 13, 256, 256, 'plain'),         # 4 chunks, 6656 tiles: 26 per workgroup
    (64, 7, 512, 480, 'mask'),           # the U-Net's data gradients: LeakyReLU-backward mask
    (64, 27, 128, 200, 'res1'),          # ragged width (last tile column 8 px), a residual source, strips cross images
    (48, 13, 256, 256, 'res2'),          # 3 chunks
    (32, 12, 256, 288, 'plain'),         # 2 chunks
    (16, 14, 256, 256, 'mask'),          # 1 chunk: every ring slot is a tile
    (16, 13, 256, 256, 'plain'),
    (64, 4, 256, 256, 'upsample'),       # nearest x2 on the fly (conv_up1 / conv_up2): 512x512 output
    (64, 3, 1040, 1008, 'nobias'),       # big images, strips end inside them
])
def test_stream_conv_equals_per_tile_kernel_bit_for_bit(cuda, cin, n, h, w, extra):
    lib = _lib.load()
    g = torch.Generator().manual_seed(cin + h + w)
    cout = 64
    x = H.CB16(torch.randn(n, cin // 16, h, w, 16, generator=g).to(torch.bfloat16).to(cuda))
    wt = (torch.randn(cout, cin, 3, 3, generator=g) * (0.7 / (cin * 9) ** 0.5)).to(cuda)
    b = None if extra == 'nobias' else (torch.randn(cout, generator=g) * 0.1).to(cuda)
    pc = H.PackedConvBF16(wt, b)
    up = extra == 'upsample'
    oh, ow = (2 * h, 2 * w) if up else (h, w)
    other = H.CB16(torch.randn(n, cout // 16, oh, ow, 16, generator=g).to(torch.bfloat16).to(cuda))
    kw = dict(act_slope=0.2, alpha=0.7, upsample=up)
    if extra == 'mask':
        kw = dict(act_slope=1.0, mask=other, mask_slope=0.2)
    elif extra == 'res1':
        kw.update(res1=other, beta1=0.3)
    elif extra == 'res2':
        kw.update(res2=other, beta2=1.0)
    outs = []
    for stream in (0, 1, 1):
        lib.sr_dev_set_conv_stream(stream)
        out = H.CB16(torch.full((n, cout // 16, oh, ow, 16), 5.0, dtype=torch.bfloat16, device=cuda))
        H.conv3x3_bf16(x, pc, out, **kw)
        torch.cuda.synchronize()
        outs.append(out.buf)
    assert bool(torch.isfinite(outs[0].float()).all())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])

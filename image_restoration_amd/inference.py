"""Single-image / folder super-resolution with an RRDBNet checkpoint on the HIP path.

The reference's inference.py serves a different model (GFPGANv1OCR, inference.py:28-40); what this script keeps is
its I/O convention: read BGR uint8, /255, BGR->RGB CHW float (img2tensor, img_util.py:9-35),
network, clamp to [0,1], RGB->BGR HWC, *255 round (tensor2img, img_util.py:38-94 with min_max=(0,1) as
sr_model.py:148).  Large frames go through the tiler (tiling.py).

    python -m image_restoration_amd.inference --input crop.png --output out.png --model_path net_g.pth \
        [--num_block 23 --num_feat 64 --tile 512 --tile_pad 16 --compute_dtype fp32|bf16]
"""
import argparse
import glob
import os

import numpy as np
import torch

from .archs import build_network
from .tiling import tiled_forward
from .utils.img_util import img2tensor, tensor2img


def imread_bgr(path):
    from PIL import Image
    rgb = np.asarray(Image.open(path).convert('RGB'))
    return np.ascontiguousarray(rgb[:, :, ::-1])


def imwrite_bgr(path, img):
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(np.ascontiguousarray(img[:, :, ::-1])).save(path)


def load_generator(args, device):
    net = build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=args.scale, num_feat=args.num_feat,
                             num_block=args.num_block, num_grow_ch=args.num_grow_ch,
                             compute_dtype=getattr(args, 'compute_dtype', 'fp32')))
    if args.model_path:
        from .utils.checkpoint import load_generator_weights
        load_generator_weights(net, args.model_path, strict=True)  # BasicSR files and official ESRGAN key names
    return net.to(device).eval()


def restore(net, img_bgr_u8, tile=0, tile_pad=16, scale=4):
    x = img2tensor(img_bgr_u8.astype(np.float32) / 255., bgr2rgb=True, float32=True).unsqueeze(0).to(next(net.parameters()).device)
    with torch.no_grad():
        y = tiled_forward(net, x, tile, tile_pad, scale) if tile and max(x.shape[2:]) > tile else net(x)
    return tensor2img(y, rgb2bgr=True, min_max=(0, 1))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--input', required=True, help='image file or folder')
    ap.add_argument('--output', required=True, help='output file or folder')
    ap.add_argument('--model_path', default=None)
    ap.add_argument('--scale', type=int, default=4)
    ap.add_argument('--num_feat', type=int, default=64)
    ap.add_argument('--num_block', type=int, default=23)
    ap.add_argument('--num_grow_ch', type=int, default=32)
    ap.add_argument('--tile', type=int, default=0)
    ap.add_argument('--tile_pad', type=int, default=16)
    ap.add_argument('--compute_dtype', choices=('fp32', 'bf16'), default='fp32',
                    help='fp32 = the reference arithmetic; bf16 = reduced-precision kernels (about 6x faster)')
    args = ap.parse_args(argv)
    net = load_generator(args, torch.device('cuda'))
    paths = sorted(glob.glob(os.path.join(args.input, '*'))) if os.path.isdir(args.input) else [args.input]
    for p in paths:
        out = restore(net, imread_bgr(p), args.tile, args.tile_pad, args.scale)
        dst = os.path.join(args.output, os.path.basename(p)) if os.path.isdir(args.input) else args.output
        imwrite_bgr(dst, out)
        print(f'{p} -> {dst} {out.shape}')


if __name__ == '__main__':
    main()

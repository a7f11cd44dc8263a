"""Single-image / folder super-resolution with an RRDBNet checkpoint on the HIP path.

The reference's inference.py serves a different model (GFPGANv1OCR, inference.py:28-40); what this script keeps is
its I/O convention: read BGR uint8, /255, BGR->RGB CHW float (img2tensor, img_util.py:9-35),
network, clamp to [0,1], RGB->BGR HWC, *255 round (tensor2img, img_util.py:38-94 with min_max=(0,1) as
sr_model.py:148).  Large frames go through the tiler (tiling.py).

    python -m image_restoration_amd.inference --input crop.png --output out.png --model_path net_g.pth \
        [--num_block 23 --num_feat 64 --tile 512 --tile_pad 16 --compute_dtype fp32|bf16]
    python -m torch.distributed.run --nproc-per-node 8 -m image_restoration_amd.inference --launcher pytorch --tile 512 ...
"""
import argparse
import glob
import os

import numpy as np
import torch

from .archs import build_network
from . import watchdog
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
    if not args.model_path:
        torch.manual_seed(0)  # no checkpoint: every rank of a sharded run must still hold the same (random) weights
    net = build_network(dict(type='RRDBNet', num_in_ch=3, num_out_ch=3, scale=args.scale, num_feat=args.num_feat,
                             num_block=args.num_block, num_grow_ch=args.num_grow_ch,
                             compute_dtype=getattr(args, 'compute_dtype', 'fp32')))
    if args.model_path:
        from .utils.checkpoint import load_generator_weights
        load_generator_weights(net, args.model_path, strict=True)  # BasicSR files and official ESRGAN key names
    return net.to(device).eval()


def restore(net, img_bgr_u8, tile=0, tile_pad=16, scale=4, rank=0, world_size=1):
    """uint8 BGR image -> uint8 BGR x`scale` image.  With world_size > 1 (one process per GPU, every rank calls this with the
    same image) the tiles are sharded over the ranks and the result is assembled on rank 0 (None elsewhere)."""
    x = img2tensor(img_bgr_u8.astype(np.float32) / 255., bgr2rgb=True, float32=True).unsqueeze(0).to(next(net.parameters()).device)
    with torch.no_grad():
        if tile and (max(x.shape[2:]) > tile or world_size > 1):
            y = tiled_forward(net, x, tile, tile_pad, scale, rank=rank, world_size=world_size)
        else:
            y = watchdog.guarded(lambda: net(x), 'restore') if rank == 0 else None   # tiled_forward guards itself
    return None if y is None else tensor2img(y, rgb2bgr=True, min_max=(0, 1))


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
    ap.add_argument('--launcher', choices=('none', 'pytorch'), default='none',
                    help="pytorch: started by torch.distributed.run, one process per GPU; the tiles of every image (--tile) are "
                         "sharded over the ranks, rank 0 writes the results")
    ap.add_argument('--dist_backend', default='nccl', help='process-group backend of --launcher pytorch (nccl = RCCL)')
    ap.add_argument('--compute_dtype', choices=('fp32', 'bf16'), default='fp32',
                    help='fp32 = the reference arithmetic; bf16 = reduced-precision kernels (about 6x faster)')
    args = ap.parse_args(argv)
    rank, world = 0, 1
    if args.launcher == 'pytorch':
        from .utils.dist_util import get_dist_info, init_dist
        init_dist('pytorch', backend=args.dist_backend)
        rank, world = get_dist_info()
        if not args.tile:
            ap.error('--launcher pytorch shards tiles: give --tile')
    net = load_generator(args, torch.device('cuda'))
    paths = sorted(glob.glob(os.path.join(args.input, '*'))) if os.path.isdir(args.input) else [args.input]
    for p in paths:
        out = restore(net, imread_bgr(p), args.tile, args.tile_pad, args.scale, rank, world)
        if out is None:
            continue
        dst = os.path.join(args.output, os.path.basename(p)) if os.path.isdir(args.input) else args.output
        imwrite_bgr(dst, out)
        print(f'{p} -> {dst} {out.shape}')
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

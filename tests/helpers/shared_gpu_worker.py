"""One process of the shared-GPU test of the fused bf16 dense block (tests/test_watchdog_gpu.py).  With ``--world 2`` it is started
twice by torch.distributed.run (gloo rendezvous; both ranks use the box's one GPU), with ``--world 1`` once, directly.  Every process
builds the same 23-block bf16 generator (seeded weights), and runs ``--reps`` forwards of the same 2 x 3 x 544 x 544 batch — two
padded cells of the 4K tiler, 578 tiles of 16 x 32 each, more than the chip has CUs — through the product's module, NOT through the
watchdog's fallback: what is tested is that the fused kernel itself makes progress on whatever share of the CUs a process gets
(tiles are claimed from a counter in dependency order, conv_bf16.hip).  Written: SHA-256 of every output, the watchdog's verdict."""
import argparse
import hashlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', required=True)
    ap.add_argument('--world', type=int, default=1)
    ap.add_argument('--reps', type=int, default=4)
    ap.add_argument('--cell', type=int, default=544)
    ap.add_argument('--num_block', type=int, default=23)
    args = ap.parse_args()
    rank = 0
    if args.world > 1:
        from image_restoration_amd.utils.dist_util import get_dist_info, init_dist
        init_dist('pytorch', backend='gloo')
        rank, world = get_dist_info()
        assert world == args.world
    import image_restoration_amd as ira
    from image_restoration_amd import _lib, watchdog
    from image_restoration_amd.utils import synth
    dev = torch.device('cuda:0')
    cfg = dict(num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=args.num_block, num_grow_ch=32)
    net = ira.build_network(dict(type='RRDBNet', compute_dtype='bf16', **cfg)).to(dev).eval()
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(0, **cfg).items()}, strict=True)
    x = torch.from_numpy(synth.uniform_input(77, (2, 3, args.cell, args.cell))).to(dev)
    with torch.no_grad():
        net(x[:, :, :64, :64])                       # packing, workspaces, module load: out of the contended section
    torch.cuda.synchronize()
    assert _lib.load().sr_chain_watchdog() == 0
    if args.world > 1:
        torch.distributed.barrier()                  # both ranks start their launches together
    digests, errors = [], []
    with torch.no_grad():
        for _ in range(args.reps):
            try:
                y = net(x)
                torch.cuda.synchronize()
                digests.append(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())
            except _lib.SrHipError as exc:           # the driver found an earlier launch's time-out
                errors.append(str(exc))
    tripped = watchdog.tripped()
    json.dump(dict(rank=rank, digests=digests, errors=errors, tripped=bool(tripped), fallbacks=watchdog.fallback_count),
              open(os.path.join(args.out, f'shared_rank{rank}_of{args.world}.json'), 'w'))
    if args.world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

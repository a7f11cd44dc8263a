"""Tile-sharded super-resolution of large frames (BASELINE config 5; SURVEY.md §8 rows a10, e).

The reference has no tiler (sr_model.py:120-129 runs the whole image in one forward), so the rule is the
build's own, Real-ESRGAN style: cut the LR frame into `tile` x `tile` cells, extend every cell by `pad` LR pixels
on each side (clipped at the frame border), super-resolve the padded cell, and paste only the cell's own
`scale*tile` centre.  Parity = the same rule with the reference's RRDBNet.forward per padded cell (golden G-k).

Cells are independent, so they shard across data-parallel ranks with no data-path collective: rank r takes cells
r, r+world, ...; cells of equal padded shape are batched into one forward; the HR crops meet on the assembling
rank by one device-side gather (the path's only exchange step).
"""
import torch

from . import watchdog


def plan_tiles(h, w, tile, pad):
    """List of cells: (y0, y1, x0, x1) of the cell and (py0, py1, px0, px1) of its padded window, LR coordinates."""
    cells = []
    for y0 in range(0, h, tile):
        for x0 in range(0, w, tile):
            y1, x1 = min(y0 + tile, h), min(x0 + tile, w)
            cells.append(((y0, y1, x0, x1), (max(y0 - pad, 0), min(y1 + pad, h), max(x0 - pad, 0), min(x1 + pad, w))))
    return cells


def _run_cells(net, img, cells, scale, max_batch):
    """Super-resolves the padded windows of `cells`, batching windows of equal shape.  Returns {cell index: HR crop}."""
    by_shape = {}
    for i, (_, (py0, py1, px0, px1)) in cells:
        by_shape.setdefault((py1 - py0, px1 - px0), []).append(i)
    lookup = dict(cells)
    out = {}
    for _, idxs in sorted(by_shape.items()):
        # equal shares instead of full batches + a remainder (18 cells at max 8: 6 + 6 + 6, not 8 + 8 + 2): a forward's dense blocks walk
        # their tiles in rounds of one per CU, so a small last batch pays a whole round for a fraction of one
        parts = -(-len(idxs) // max_batch)
        share = -(-len(idxs) // parts)
        for s in range(0, len(idxs), share):
            chunk = idxs[s:s + share]
            batch = torch.cat([img[:, :, lookup[i][1][0]:lookup[i][1][1], lookup[i][1][2]:lookup[i][1][3]] for i in chunk], 0)
            with torch.no_grad():
                sr = net(batch.contiguous())
            for b, i in enumerate(chunk):
                (y0, y1, x0, x1), (py0, _, px0, _) = lookup[i]
                oy, ox = (y0 - py0) * scale, (x0 - px0) * scale
                out[i] = sr[b:b + 1, :, oy:oy + (y1 - y0) * scale, ox:ox + (x1 - x0) * scale]
    return out


def quantise_u8(t):
    """tensor2img's output convention (img_util.py:98-147 of the reference): clamp to [0, 1], x255, round, uint8."""
    return (t.clamp(0, 1) * 255.0).round().to(torch.uint8)


def _gather_crops(crops, cells, world_size, rank, dst, group, n_out, scale, dtype, device):
    """The exchange step of the sharded tiler: every rank flattens its HR crops (cell-index order) into one device buffer
    and rank `dst` receives them with ONE `gather` (RCCL over xGMI on GPUs; CPU tensors under gloo).  Sizes follow from the
    tile plan, which every rank knows, so no size exchange is needed; buffers are padded to the largest share."""
    import torch.distributed as dist

    def numel(i):
        y0, y1, x0, x1 = cells[i][1][0]
        return n_out * (y1 - y0) * scale * (x1 - x0) * scale
    owned = [[i for i, _ in cells if i % world_size == r] for r in range(world_size)]
    cap = max(sum(numel(i) for i in o) for o in owned)
    on_host = dist.get_backend(group) == 'gloo'
    buf_dev = torch.device('cpu') if on_host else device
    send = torch.zeros(cap, dtype=dtype, device=device)
    off = 0
    for i in owned[rank]:
        send[off:off + numel(i)] = crops[i].reshape(-1)
        off += numel(i)
    send = send.to(buf_dev)
    recv = [torch.empty(cap, dtype=dtype, device=buf_dev) for _ in range(world_size)] if rank == dst else None
    dist.gather(send, recv, dst=dst, group=group)
    if rank != dst:
        return None
    out = {}
    for r in range(world_size):
        off = 0
        part = recv[r].to(device)
        for i in owned[r]:
            y0, y1, x0, x1 = cells[i][1][0]
            out[i] = part[off:off + numel(i)].view(1, n_out, (y1 - y0) * scale, (x1 - x0) * scale)
            off += numel(i)
    return out


def tiled_forward(net, img, tile=512, pad=16, scale=4, max_batch=8, rank=0, world_size=1, group=None, dst=0,
                  out_dtype=torch.float32):
    """img [1, C, H, W] on the HIP device -> [1, C_out, scale*H, scale*W] on rank `dst` (None elsewhere when
    world_size > 1).  With world_size > 1 every rank must call this with the same image.  out_dtype torch.uint8 applies the
    output convention (clamp, x255, round) on the producing rank, which also cuts the exchange to a quarter."""
    assert img.dim() == 4 and img.size(0) == 1 and out_dtype in (torch.float32, torch.uint8)
    h, w = img.shape[2:]
    cells = list(enumerate(plan_tiles(h, w, tile, pad)))
    mine = [c for c in cells if c[0] % world_size == rank]
    # the crops are about to leave the device (gather, uint8 copy, the caller's imwrite): a dense-block launch that timed out
    # must not hand over invalid pixels (watchdog.py; the cells are repeated on the chain launch in this process)
    if img.is_cuda:
        crops = watchdog.guarded(lambda: _run_cells(net, img, mine, scale, max_batch), 'tiled_forward')
    else:
        crops = _run_cells(net, img, mine, scale, max_batch)
    if out_dtype == torch.uint8:
        crops = {i: quantise_u8(c) for i, c in crops.items()}
    n_out = next(iter(crops.values())).size(1) if crops else getattr(net, 'num_out_ch', img.size(1))
    if world_size > 1:
        crops = _gather_crops(crops, cells, world_size, rank, dst, group, n_out, scale, out_dtype, img.device)
        if crops is None:
            return None
    out = torch.empty((1, n_out, h * scale, w * scale), dtype=out_dtype, device=img.device)
    for i, ((y0, y1, x0, x1), _) in cells:
        out[:, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = crops[i]
    return out

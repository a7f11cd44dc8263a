"""Tile-sharded super-resolution of large frames (BASELINE config 5; SURVEY.md §8 rows a10, e).

The reference has no tiler (sr_model.py:120-129 runs the whole image in one forward), so the rule is the
build's own, Real-ESRGAN style: cut the LR frame into `tile` x `tile` cells, extend every cell by `pad` LR pixels
on each side (clipped at the frame border), super-resolve the padded cell, and paste only the cell's own
`scale*tile` centre.  Parity = the same rule with the reference's RRDBNet.forward per padded cell (golden G-k).

Cells are independent, so they shard across data-parallel ranks with no data-path collective: rank r takes cells
r, r+world, ...; cells of equal padded shape are batched into one forward; the HR crops meet on the assembling
rank by one gather (or stay sharded if the caller wants them in place).
"""
import torch


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
        for s in range(0, len(idxs), max_batch):
            chunk = idxs[s:s + max_batch]
            batch = torch.cat([img[:, :, lookup[i][1][0]:lookup[i][1][1], lookup[i][1][2]:lookup[i][1][3]] for i in chunk], 0)
            with torch.no_grad():
                sr = net(batch.contiguous())
            for b, i in enumerate(chunk):
                (y0, y1, x0, x1), (py0, _, px0, _) = lookup[i]
                oy, ox = (y0 - py0) * scale, (x0 - px0) * scale
                out[i] = sr[b:b + 1, :, oy:oy + (y1 - y0) * scale, ox:ox + (x1 - x0) * scale]
    return out


def tiled_forward(net, img, tile=512, pad=16, scale=4, max_batch=8, rank=0, world_size=1, group=None, dst=0):
    """img [1, C, H, W] on the HIP device -> [1, C_out, scale*H, scale*W] on rank `dst` (None elsewhere when
    world_size > 1).  With world_size > 1 every rank must call this with the same image."""
    assert img.dim() == 4 and img.size(0) == 1
    h, w = img.shape[2:]
    cells = list(enumerate(plan_tiles(h, w, tile, pad)))
    mine = [c for c in cells if c[0] % world_size == rank]
    crops = _run_cells(net, img, mine, scale, max_batch)
    if world_size > 1:
        import torch.distributed as dist
        gathered = [None] * world_size if rank == dst else None
        dist.gather_object({i: c.cpu() for i, c in crops.items()}, gathered, dst=dst, group=group)
        if rank != dst:
            return None
        crops = {}
        for part in gathered:
            crops.update({i: c.to(img.device) for i, c in part.items()})
    n_out = next(iter(crops.values())).size(1)
    out = torch.empty((1, n_out, h * scale, w * scale), dtype=torch.float32, device=img.device)
    for i, ((y0, y1, x0, x1), _) in cells:
        out[:, :, y0 * scale:y1 * scale, x0 * scale:x1 * scale] = crops[i]
    return out

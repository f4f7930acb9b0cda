"""Image-tile partition across ranks and the gather of the framebuffer to rank 0.

This is the multi-GPU form of the reference's only parallelism, rows strided over 16
threads (src/flyscene.cpp:418-430,460): here tile g of the row-major tile grid belongs to
rank g % world.  Each rank renders its tiles into a compact buffer
(`rtx_render_tiles_dev`); one `torch.distributed.gather` per output moves them to rank 0
(RCCL send/recv over xGMI when the backend is nccl, direct peer-to-root, no ring), where
`assemble` turns the (rank, local tile) order back into the row-major image.
"""
import torch


def grid(width, height, ts):
    return (width + ts - 1) // ts, (height + ts - 1) // ts


def local_tiles(width, height, ts, rank, world):
    tx, ty = grid(width, height, ts)
    total = tx * ty
    return max(0, (total - rank + world - 1) // world)


def max_local_tiles(width, height, ts, world):
    return local_tiles(width, height, ts, 0, world)


def assemble(parts, width, height, ts, channels):
    """parts: tensor [world, kmax*ts*ts, channels] (rank-major, local tile k of rank p is
    global tile k*world + p).  Returns the [height, width, channels] image."""
    world = parts.shape[0]
    tx, ty = grid(width, height, ts)
    kmax = parts.shape[1] // (ts * ts)
    t = parts.reshape(world, kmax, ts, ts, channels).transpose(0, 1).reshape(kmax * world, ts, ts, channels)
    t = t[: tx * ty].reshape(ty, tx, ts, ts, channels).permute(0, 2, 1, 3, 4).reshape(ty * ts, tx * ts, channels)
    return t[:height, :width]


def gather_to_root(local, world, rank, dist=None, out=None):
    """local: [kmax*ts*ts, channels] on every rank (same shape).  Returns [world, ...] on rank 0, None elsewhere.
    out: optional preallocated [world, ...] tensor on rank 0 (the gather then lands in place: no per-frame
    allocation, no stacking copy)."""
    if world == 1:
        return local.unsqueeze(0)
    if rank != 0:
        dist.gather(local, None, dst=0)
        return None
    if out is None:
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
    dist.gather(local, [out[i] for i in range(world)], dst=0)
    return out

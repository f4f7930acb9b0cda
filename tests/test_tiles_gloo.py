"""The N>1 path on CPU: tile partition + torch.distributed gather (gloo, world_size 2 and 3)
reassembles exactly the single-process frame.  The renderer here is a pure index function
(every pixel gets a value computed from its coordinates), standing in for rtx_render_tiles_dev,
whose own tile layout is checked against full frames on the GPU (test_gpu_parity.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_project_amd import tiles


def fake_render_tiles(W, H, ts, part, nparts, kmax):
    """What rtx_render_tiles_dev writes: local tile k = global tile k*nparts+part, row-major inside."""
    tx, ty = tiles.grid(W, H, ts)
    out = torch.zeros((kmax * ts * ts, 3), dtype=torch.float32)
    n = tiles.local_tiles(W, H, ts, part, nparts)
    for k in range(n):
        g = k * nparts + part
        gx, gy = g % tx, g // tx
        for iy in range(ts):
            y = gy * ts + iy
            for ix in range(ts):
                x = gx * ts + ix
                if x < W and y < H:
                    out[k * ts * ts + iy * ts + ix] = torch.tensor([x, y, x * 1000 + y], dtype=torch.float32)
    return out


def expected(W, H):
    ys, xs = np.mgrid[0:H, 0:W]
    return np.stack([xs, ys, xs * 1000 + ys], -1).astype(np.float32)


def _worker(rank, world, port, W, H, ts, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kmax = tiles.max_local_tiles(W, H, ts, world)
    local = fake_render_tiles(W, H, ts, rank, world, kmax)
    parts = tiles.gather_to_root(local, world, rank, dist)
    # the bench's form: preallocated destination on rank 0, reused across frames
    pre = torch.full((world,) + tuple(local.shape), -7.0) if rank == 0 else None
    parts2 = None
    for _ in range(2):
        parts2 = tiles.gather_to_root(local, world, rank, dist, pre)
    # batches (bench.py with --batch): nf frames' tiles of a rank one after the other, gathered in ONE exchange into a
    # slice of a buffer sized for the largest batch; frame f of every part is then reassembled on its own
    nf, cap = 2, 3
    npix = local.shape[0]
    batch = torch.cat([local + 100000.0 * f for f in range(nf)])
    big = torch.full((world, cap * npix, 3), -7.0) if rank == 0 else None
    got = tiles.gather_to_root(batch, world, rank, dist, big[:, : nf * npix] if rank == 0 else None)
    batch_ok = True
    if rank == 0:
        for f in range(nf):
            imgf = tiles.assemble(got[:, f * npix:(f + 1) * npix], W, H, ts, 3).numpy()
            batch_ok = batch_ok and bool(np.array_equal(imgf, expected(W, H) + np.float32(100000.0 * f)))
        batch_ok = batch_ok and bool((big[:, nf * npix:] == -7.0).all())
    if rank == 0:
        ret["batch_ok"] = batch_ok
        img = tiles.assemble(parts, W, H, ts, 3).numpy()
        img2 = tiles.assemble(parts2, W, H, ts, 3).numpy()
        ret["ok"] = bool(np.array_equal(img, expected(W, H))) and bool(np.array_equal(img2, expected(W, H))) and parts2 is pre
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,W,H,ts", [(2, 80, 48, 16), (3, 70, 50, 16), (2, 33, 17, 32)])
def test_gather_reassembles_frame(world, W, H, ts):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), W, H, ts, ret), nprocs=world, join=True)
    assert ret.get("ok") is True
    assert ret.get("batch_ok") is True


def test_single_rank_assemble_and_counts():
    W, H, ts = 50, 35, 16
    kmax = tiles.max_local_tiles(W, H, ts, 1)
    assert kmax == 4 * 3
    img = tiles.assemble(fake_render_tiles(W, H, ts, 0, 1, kmax).unsqueeze(0), W, H, ts, 3).numpy()
    assert np.array_equal(img, expected(W, H))
    # ragged split: 12 tiles over 5 ranks -> 3,3,2,2,2
    assert [tiles.local_tiles(W, H, ts, r, 5) for r in range(5)] == [3, 3, 2, 2, 2]


def test_tiles_count_matches_c_abi():
    from raytracing_project_amd import api
    for (W, H, ts, n) in [(3840, 2160, 32, 8), (50, 35, 16, 5), (1, 1, 16, 3)]:
        for r in range(n):
            assert api.tiles_count((0, 0, W, H), ts, r, n) == tiles.local_tiles(W, H, ts, r, n)
    with pytest.raises(api.RtxError):
        api.tiles_count((0, 0, 10, 10), 20, 0, 1)

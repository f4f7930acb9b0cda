#!/usr/bin/env python3
"""Per-kernel times of rank 0's share of an N-rank frame on one GPU (strong-scaling proxy, bench.py workload 0)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from raytracing_project_amd import api, scenes, tiles

name = sys.argv[1] if len(sys.argv) > 1 else "soup1M_4k_primary_shadow"
TILE = int(os.environ.get("TILE", "32"))
wl = bench.build_workload(name, api, scenes)
W, H = wl["W"], wl["H"]
scene = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
scene.set_lights(wl["lights"]); scene.commit(0)
opts = api.Options(*wl["opts"])
cam = api.make_camera(wl["view"], W, H)
rect = (0, 0, W, H)
dev = torch.device("cuda", 0)
for nparts in (1, 2, 4, 8):
    kmax = tiles.max_local_tiles(W, H, TILE, nparts)
    rgb = torch.zeros((kmax * TILE * TILE, 3), dtype=torch.float32, device=dev)
    face = torch.full((kmax * TILE * TILE, 1), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for _ in range(9):
        scene.render_tiles_dev(cam, opts, rect, TILE, 0, nparts, rgb.data_ptr(), None, face.data_ptr())
    acc = np.zeros(4)
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        scene.render_tiles_dev(cam, opts, rect, TILE, 0, nparts, rgb.data_ptr(), None, face.data_ptr())
        st = scene.stats()
        acc += [st.ms_trace_camera, st.ms_trace - st.ms_trace_camera, st.ms_total, st.rays_primary + st.rays_shadow + st.rays_reflection]
    dt = (time.perf_counter() - t0) / K * 1e3
    acc /= K
    print("nparts %d: camera %.3f ms, queue %.3f ms, all kernels %.3f ms, wall %.3f ms, rays %.0f" % (nparts, acc[0], acc[1], acc[2], dt, acc[3]), flush=True)

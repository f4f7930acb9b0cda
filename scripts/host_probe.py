#!/usr/bin/env python3
"""Where the wall time of a small share goes: host time inside the enqueue call vs waiting in render_finish (part 0 of N)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from raytracing_project_amd import api, scenes, tiles

name = sys.argv[1] if len(sys.argv) > 1 else "soup1M_4k_primary_shadow"
TILE = 32
wl = bench.build_workload(name, api, scenes)
W, H = wl["W"], wl["H"]
scene = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
scene.set_lights(wl["lights"]); scene.commit(0)
opts = api.Options(*wl["opts"])
cam = api.make_camera(wl["view"], W, H)
rect = (0, 0, W, H)
dev = torch.device("cuda", 0)
for nparts in (1, 8, 16):
    kmax = tiles.max_local_tiles(W, H, TILE, nparts)
    rgb = torch.zeros((kmax * TILE * TILE, 3), dtype=torch.float32, device=dev)
    face = torch.full((kmax * TILE * TILE, 1), -1, dtype=torch.int32, device=dev)
    for inflight in (1, 2, 3, 4):
        def run(n):
            te = tf = 0.0
            flying = 0
            for k in range(n):
                t0 = time.perf_counter()
                scene.render_tiles_dev_enqueue(cam, opts, rect, TILE, 0, nparts, rgb.data_ptr(), None, face.data_ptr())
                t1 = time.perf_counter(); te += t1 - t0
                flying += 1
                if flying >= inflight:
                    scene.render_finish(); flying -= 1
                    tf += time.perf_counter() - t1
            t1 = time.perf_counter()
            for _ in range(flying):
                scene.render_finish()
            tf += time.perf_counter() - t1
            return te, tf
        run(10)
        torch.cuda.synchronize()
        K = 40
        t0 = time.perf_counter()
        te, tf = run(K)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("nparts %2d inflight %d: wall %.3f ms/frame, host in enqueue %.3f, host in finish %.3f" % (nparts, inflight, dt / K * 1e3, te / K * 1e3, tf / K * 1e3), flush=True)

print("batches (rtx_render_tiles_batch_dev_enqueue): ms per FRAME", flush=True)
for nparts in (1, 2, 4, 8):
    kmax = tiles.max_local_tiles(W, H, TILE, nparts)
    for B in (1, 2, 4, 8):
        rgb = torch.zeros((B * kmax * TILE * TILE, 3), dtype=torch.float32, device=dev)
        face = torch.full((B * kmax * TILE * TILE, 1), -1, dtype=torch.int32, device=dev)
        for inflight in (1, 2, 3):
            def run(n):
                flying = 0
                for k in range(n):
                    scene.render_tiles_batch_dev([cam] * B, opts, rect, TILE, 0, nparts, rgb.data_ptr(), None, face.data_ptr(), enqueue=True)
                    flying += 1
                    if flying >= inflight:
                        scene.render_finish(); flying -= 1
                for _ in range(flying):
                    scene.render_finish()
            run(10)
            torch.cuda.synchronize()
            K = 24
            t0 = time.perf_counter()
            run(K)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print("nparts %2d batch %d inflight %d: %.3f ms/frame" % (nparts, B, inflight, dt / (K * B) * 1e3), flush=True)

#!/usr/bin/env python3
"""Frames of rank 0's share of an N-rank run, D in flight -- to be run under rocprofv3 --kernel-trace; or, with
--analyze <trace.db>, how the kernels of the traced run overlapped (kernel-seconds per wall second)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "--analyze":
    import sqlite3
    c = sqlite3.connect(sys.argv[2])
    rows = c.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall() if False else None
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    print("columns:", cols)
    q = "select name, start, end" + (", queue_id" if "queue_id" in cols else ", 0") + " from kernels order by start"
    rows = c.execute(q).fetchall()
    rows = [r for r in rows if "rtx::" in r[0]]
    rows = rows[len(rows) // 2:]                  # the second half of the run: steady state
    t0, t1 = min(r[1] for r in rows), max(r[2] for r in rows)
    busy = sum(r[2] - r[1] for r in rows)
    print("%d kernels over %.3f ms wall, %.3f ms of kernel time -> %.2f kernels running on average" % (len(rows), (t1 - t0) / 1e6, busy / 1e6, busy / (t1 - t0)))
    import collections
    per = collections.defaultdict(list)
    for n, a, b, qid in rows:
        per[n.split("(")[0][:40]].append((b - a) / 1e3)
    for n, v in per.items():
        print("  %-42s n=%4d avg %.1f us" % (n, len(v), sum(v) / len(v)))
    # time with >= 2 traversal kernels running
    ev = []
    for n, a, b, qid in rows:
        if "k_trace" in n:
            ev += [(a, 1), (b, -1)]
    ev.sort()
    cur, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[cur] += t - last
        cur += d; last = t
    tot = sum(hist.values())
    print("  traversal kernels running at once:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
    print("  queues used:", sorted({r[3] for r in rows}))
    sys.exit(0)
import numpy as np
import torch
import bench
from raytracing_project_amd import api, scenes, tiles
nparts = int(os.environ.get("NPARTS", "8")); D = int(os.environ.get("INFLIGHT", "4")); frames = int(os.environ.get("FRAMES", "60"))
TILE = 32
wl = bench.build_workload("soup1M_4k_primary_shadow", api, scenes)
W, H = wl["W"], wl["H"]
scene = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
scene.set_lights(wl["lights"]); scene.commit(0)
opts = api.Options(*wl["opts"]); cam = api.make_camera(wl["view"], W, H); rect = (0, 0, W, H)
kmax = tiles.max_local_tiles(W, H, TILE, nparts)
bufs = [(torch.zeros((kmax * TILE * TILE, 3), dtype=torch.float32, device="cuda"), torch.zeros((kmax * TILE * TILE, 1), dtype=torch.int32, device="cuda")) for _ in range(D + 1)]
torch.cuda.synchronize()
def run(n):
    fly = 0
    for k in range(n):
        rgb, face = bufs[k % (D + 1)]
        scene.render_tiles_dev_enqueue(cam, opts, rect, TILE, 0, nparts, rgb.data_ptr(), None, face.data_ptr())
        fly += 1
        if fly >= D:
            scene.render_finish(); fly -= 1
    for _ in range(fly):
        scene.render_finish()
run(10)
t0 = time.perf_counter(); run(frames); dt = time.perf_counter() - t0
print("nparts %d, %d in flight: %.3f ms per frame" % (nparts, D, dt / frames * 1e3))

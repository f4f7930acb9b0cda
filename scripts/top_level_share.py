"""How many of a primary ray's group steps lie in the top D levels of the hierarchy (CPU only: the host builder, the exported
groups, a Python replay of the traversal pruned at the ray's final hit distance)?  The steps above a frontier at depth D are what a
per-tile entry-point search could take off every ray of a coherent tile.  Usage: python scripts/top_level_share.py [workload] [rays]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from raytracing_project_amd import api, scenes
    from oracle import oracle_api as oa
    oa.build()
    name = sys.argv[1] if len(sys.argv) > 1 else "soup1M_4k_primary_shadow"
    nr = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    wl = bench.build_workload(name, api, scenes)
    s = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
    s.build()
    info = s.info()
    nodes, order = s.bvh()
    tris = s.tris()
    q = nodes.reshape(-1, 4, 8)
    lo = q[:, :, 0:3].view(np.float32); hi = q[:, :, 4:7].view(np.float32)
    ref = q[:, :, 3]; cnt = q[:, :, 7]
    W, H = wl["W"], wl["H"]
    cam = oa.make_camera(wl["view"], W, H)
    o, d1 = oa.camera_rays(cam, (0, 0, W, H))
    rng = np.random.default_rng(1)
    pick = rng.choice(len(o), nr, replace=False)
    o, d1 = o[pick], d1[pick]
    v = wl["verts"]; rlo = v.min(0); rhi = np.maximum(v.max(0), np.float32(1.17549435e-38))
    t, f, nv, tt, gs, tf = oa.bvh_trace(nodes, tris, order, rlo, rhi, o, d1, threads=8, nwild=info.nwild, layout_counts=True)
    d = d1 / np.linalg.norm(d1, axis=1, keepdims=True)
    hist = np.zeros(40, np.int64)
    total = 0
    entering = 0
    for i in range(nr):
        if gs[i] == 0:
            continue
        entering += 1
        lim = t[i] * 1.0001 if np.isfinite(t[i]) else np.inf
        inv = 1.0 / np.where(d[i] == 0, 1e-30, d[i])
        stack = [(0, 1)]
        while stack:
            g, depth = stack.pop()
            hist[depth] += 1; total += 1
            t0 = (lo[g] - o[i]) * inv; t1 = (hi[g] - o[i]) * inv
            tn = np.minimum(t0, t1).max(1); tfar = np.maximum(t0, t1).min(1)
            hitc = (tn <= np.minimum(tfar, lim)) & (tfar >= 0) & ~((ref[g] == 0xFFFFFFFF) & (cnt[g] == 0))
            for k in np.argsort(-tn):
                if hitc[k] and cnt[g, k] == 0:
                    stack.append((int(ref[g, k]), depth + 1))
    print("%s: %d of %d sampled rays enter the hierarchy; replayed group steps per entering ray %.1f (the mirror counts %.1f)" % (
        name, entering, nr, total / max(1, entering), gs[gs > 0].mean()))
    cum = np.cumsum(hist)
    for D in range(1, 9):
        print("  steps in the top %d levels: %.1f per ray = %.0f %% of all" % (D, cum[D] / max(1, entering), 100.0 * cum[D] / max(1, total)))


if __name__ == "__main__":
    main()

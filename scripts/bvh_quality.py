#!/usr/bin/env python3
"""Group steps / triangle tests per primary ray of a workload under the builder that is loaded (CPU only: the host
builder + the oracle's mirror traversal of the exported BVH).  Compare builder variants through their environment hooks,
e.g.  RTX_COLLAPSE=fixed python3 scripts/bvh_quality.py hall262k_1080p_reflect3"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from oracle import oracle_api as oa
from raytracing_project_amd import api, scenes

name = sys.argv[1] if len(sys.argv) > 1 else "soup1M_4k_primary_shadow"
stride = int(sys.argv[2]) if len(sys.argv) > 2 else 64
wl = bench.build_workload(name, api, scenes)
W, H = wl["W"], wl["H"]
scene = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
scene.build()
info = scene.info()
nodes, order = scene.bvh()
tris = scene.tris()
_, osc = bench.oracle_scene(wl)
lo, hi = osc.root_box()
ocam = oa.make_camera(wl["view"], W, H)
o, d1 = oa.camera_rays(ocam, (0, 0, W, H))
pick = np.arange(0, W * H, stride)
t, f, nv, tt, gs, tf = oa.bvh_trace(nodes, tris, order, lo, hi, o[pick], d1[pick], threads=8, nwild=info.nwild, layout_counts=True)
ent = gs > 0
print("%s: groups %d (slots %d, %.2f per group), depth %d, build %.0f ms | per ray entering the BVH: group steps %.2f, child slots %.2f, triangles %.2f"
      % (name, info.nnodes // 4 if hasattr(info, "nnodes") else -1, info.nslots, info.nslots / max(info.nnodes / 4, 1), info.max_depth, info.build_ms,
         gs[ent].mean(), nv[ent].mean(), tt[ent].mean()))

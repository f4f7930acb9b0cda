import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, bench
from raytracing_project_amd import api, scenes
wl = bench.build_workload("soup1M_4k_primary_shadow", api, scenes)
W, H = wl["W"], wl["H"]
scene = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
scene.set_lights(wl["lights"]); scene.commit(0)
opts = api.Options(*wl["opts"]); cam = api.make_camera(wl["view"], W, H)
for want_hits in (True, False):
    for _ in range(4):
        scene.render(cam, opts, (0, 0, W, H), want_hits=want_hits)
    N = 8
    t0 = time.perf_counter()
    for _ in range(N):
        scene.render(cam, opts, (0, 0, W, H), want_hits=want_hits)
    dt = (time.perf_counter() - t0) / N
    st = scene.stats()
    rays = st.rays_primary + st.rays_shadow + st.rays_reflection
    print("PCIE host-pointer rtx_render want_hits=%s: %.2f ms per frame, %.0f Mrays/s (kernels %.2f ms)" % (want_hits, dt * 1e3, rays / dt / 1e6, st.ms_total))

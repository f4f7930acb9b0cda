"""One frame as K horizontal bands enqueued side by side (rtx_render_dev_enqueue on sub-rectangles, one call slot each) against the
single call: does overlapping the bands' launches hide the drain of a lone frame's persistent kernels?  Usage: python scripts/split_frame_probe.py [workload]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from raytracing_project_amd import api, scenes
    name = sys.argv[1] if len(sys.argv) > 1 else "soup1M_4k_primary_shadow"
    wl = bench.build_workload(name, api, scenes)
    W, H = wl["W"], wl["H"]
    s = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
    s.set_lights(wl["lights"]); s.commit(0)
    opts = api.Options(*wl["opts"])
    rgb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    ref = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    cam = lambda k: api.make_camera(bench.frame_view(wl, scenes, k), W, H)
    for k in range(10):
        s.render_dev(cam(k), opts, (0, 0, W, H), ref.data_ptr(), None, None)
    for K in (1, 2, 3, 4):
        bands = [(0, H * i // K, W, H * (i + 1) // K) for i in range(K)]
        def frame(k):
            c = cam(k)
            for b in bands:
                s.render_dev_enqueue(c, opts, b, rgb.data_ptr() + b[1] * W * 12, None, None)
            for _ in bands:
                s.render_finish()
        for k in range(24):
            frame(k)                       # tuner trials of the band's size class
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(32):
                frame(k)
            torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) / 32 * 1e3)
        s.render_dev(cam(31), opts, (0, 0, W, H), ref.data_ptr(), None, None)
        same = bool(torch.equal(rgb.view(torch.int32), ref.view(torch.int32)))
        print("%s: %d band(s): %.3f ms per frame (min %.3f), equal to the single call: %s" % (name, K, float(np.median(ts)), min(ts), same), flush=True)
    s.close()


if __name__ == "__main__":
    main()

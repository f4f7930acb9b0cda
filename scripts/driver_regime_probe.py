"""The driver's command (--steps 20 --warmup 5) seen from rank 0 of N: K = 20 frames per timed repeat, rendered as calls of B frames with
`inflight` calls side by side (part 0 of N, no gather).  Which (B, inflight) suits so short a timed region?
Usage: python scripts/driver_regime_probe.py [K]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import torch
    from raytracing_project_amd import api, scenes, tiles
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    wl = bench.build_workload("soup1M_4k_primary_shadow", api, scenes)
    W, H = wl["W"], wl["H"]
    s = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
    s.set_lights(wl["lights"]); s.commit(0)
    opts = api.Options(*wl["opts"])
    rect = (0, 0, W, H)
    dev = torch.device("cuda", 0)
    cache = {}

    def cam_of(k):                       # (camera records are input: made once, before any clock starts, as in bench.py)
        if k not in cache:
            cache[k] = api.make_camera(bench.frame_view(wl, scenes, k, False), W, H)
        return cache[k]
    base = {}
    for nparts in (1, 8):
        kmax = tiles.max_local_tiles(W, H, bench.TILE, nparts)
        npl = kmax * bench.TILE * bench.TILE if nparts > 1 else W * H
        for (B, inflight) in ((8, 2), (10, 2), (5, 4), (7, 3), (20, 1), (4, 4), (5, 2), (3, 4)):
            if nparts == 1 and B > 10:
                continue
            rgb = [torch.zeros((B * npl, 3), dtype=torch.float32, device=dev) for _ in range(inflight + 1)]
            torch.cuda.synchronize()

            def run(n, k):
                flying, left, j = 0, n, 0
                ncalls = (n + B - 1) // B
                while left > 0:
                    nf = (left + ncalls - 1) // ncalls; ncalls -= 1
                    cams = [cam_of(k + f) for f in range(nf)]
                    k += nf; left -= nf
                    s.render_tiles_batch_dev(cams, opts, rect, bench.TILE if nparts > 1 else 0, 0, nparts, rgb[j % len(rgb)].data_ptr(), None, None, enqueue=True)
                    j += 1; flying += 1
                    if flying >= inflight:
                        s.render_finish(); flying -= 1
                while flying:
                    s.render_finish(); flying -= 1
            run(7 * B, 0)
            run(K, 0)
            t = []
            for rep in range(7):
                torch.cuda.synchronize()
                t0 = time.perf_counter(); run(K, 0); torch.cuda.synchronize()
                t.append((time.perf_counter() - t0) / K * 1e3)
            t.sort()
            med = t[len(t) // 2]
            if nparts == 1:
                base[(B, inflight)] = med
            print(json.dumps({"ranks": nparts, "K": K, "frames_per_call": B, "inflight": inflight, "ms_per_frame_med": round(med, 4), "min": round(t[0], 4),
                              "speedup_vs_N1_best": round(min(base.values()) / med, 2) if nparts > 1 else None}), flush=True)
            del rgb
    s.close()


if __name__ == "__main__":
    main()

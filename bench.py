#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + reflection) of the HIP ray-scene
intersection path, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one frame of the workload: every rank renders its image tiles
(rtx_render_tiles_dev, scene replicated per GPU), then the colour image is gathered to rank 0
(RCCL) and put back into row-major order.  value = rays traced by all ranks in the K timed frames / wall
time (max over ranks).  Total work is fixed as N grows: "strong" scaling.

The JSON line also carries
  roofline      for the dominant traversal kernel: algorithmic bytes (SURVEY 8(d):
                32 B per node visited + 36 B per triangle tested + io per ray, counted by the
                CPU mirror traversal of the same BVH on a strided sample of the same rays)
                divided by that kernel's HIP-event duration, against 8 TB/s;
  cpu_baseline  the CPU oracle (brute-force reference path) timed on this host on a bounded
                sample of the same rays.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
TILE = 32


def build_workload(name, api, scenes):
    """-> dict(verts, idx, tri_mat, mats (rtx), omats (kwargs for the oracle), lights, view, W, H, opts, spp)"""
    mk = dict
    if name == "soup1M_4k_primary_shadow":
        verts, idx = scenes.soup(1000000, seed=0x5EED0001, h=0.02)
        m = mk(ka=(0, 0, 0), kd=(.7, .7, .7), ks=(.2, .2, .2), ns=10.0, ni=0.0, d=1.0, illum=2)
        return dict(verts=verts, idx=idx, tri_mat=np.zeros(len(idx), np.int32), mats=[m], lights=[[-2.0, 2.0, 2.0]],
                    view=scenes.look_at_view((0, 0, 3), (0, 0, 0)), W=3840, H=2160, opts=(0, 1, 0, 0, 1, 1), spp=1)
    if name == "soup5M_4k_16l_16spp":
        verts, idx = scenes.soup(5000000, seed=0x5EED0005, h=0.012)
        m = mk(ka=(0, 0, 0), kd=(.7, .7, .7), ks=(.2, .2, .2), ns=10.0, ni=0.0, d=1.0, illum=2)
        lights = [[3 * np.cos(2 * np.pi * k / 16), 2.0, 3 * np.sin(2 * np.pi * k / 16)] for k in range(16)]
        return dict(verts=verts, idx=idx, tri_mat=np.zeros(len(idx), np.int32), mats=[m], lights=lights,
                    view=scenes.look_at_view((0, 0, 3), (0, 0, 0)), W=3840, H=2160, opts=(0, 1, 0, 0, 1, 1), spp=16)
    if name == "bunny70k_1080p_primary_shadow":
        # BASELINE config 2: the reference's bundled resources/models/bunny.ply (35,947 verts / 69,451
        # faces), committed flattened as a fixture because /root/reference does not travel
        z = np.load(os.path.join(ROOT, "tests", "golden", "scene_bunny.npz"))
        verts, idx = z["verts"], z["idx"]
        lo, hi = verts.min(0), verts.max(0)
        c, r = (lo + hi) / 2, float(np.linalg.norm(hi - lo) / 2)
        m = mk()   # Tucano Mtl() defaults
        return dict(verts=verts, idx=idx, tri_mat=np.zeros(len(idx), np.int32), mats=[m],
                    lights=[list(c + r * np.array([-1.0, 1.0, 1.0]))],
                    view=scenes.look_at_view(c + np.array([0, 0, 2 * r]), c), W=1920, H=1080, opts=(0, 1, 0, 0, 1, 1), spp=1)
    if name == "hall262k_1080p_reflect3":
        verts, idx, tm = scenes.hall()
        wall = mk(ka=(.1, .1, .1), kd=(.6, .55, .5), ks=(.3, .3, .3), ns=20.0, ni=1.0, d=0.5, illum=2)
        col = mk(ka=(.1, .1, .1), kd=(.5, .6, .7), ks=(.6, .6, .6), ns=40.0, ni=1.0, d=0.5, illum=2)
        return dict(verts=verts, idx=idx, tri_mat=tm, mats=[wall, col], lights=[[0.0, 2.7, 0.0]],
                    view=scenes.look_at_view((0.3, 1.6, 7.0), (0.0, 1.2, 0.0)), W=1920, H=1080, opts=(3, 1, 0, 0, 1, 1), spp=1)
    if name == "planeobj_512_primary_shadow":
        z = np.load(os.path.join(ROOT, "tests", "golden", "scene_planeobj.npz"))
        mats = [mk(ka=tuple(m[0:3]), kd=tuple(m[3:6]), ks=tuple(m[6:9]), ns=float(m[9]), ni=float(m[10]), d=float(m[11]),
                   illum=int(m[12])) for m in z["mats"]]
        return dict(verts=z["verts"], idx=z["idx"], tri_mat=z["tri_mat"], mats=mats, lights=[[-1.0, 1.0, 1.0]],
                    view=scenes.default_view(), W=512, H=512, opts=(0, 1, 0, 0, 1, 1), spp=1)
    raise SystemExit("unknown workload %s" % name)


WORKLOADS = ["soup1M_4k_primary_shadow", "bunny70k_1080p_primary_shadow", "hall262k_1080p_reflect3",
             "soup5M_4k_16l_16spp", "planeobj_512_primary_shadow"]


def spp_offsets(spp):
    if spp == 1:
        return [(0.0, 0.0)]          # the reference's top-left sample
    n = int(round(spp ** 0.5))
    return [((a + .5) / n, (b + .5) / n) for b in range(n) for a in range(n)]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, one fresh process per GPU,
    BEFORE this process has made any GPU call (it never does: it only waits), the way
    `python -m torch.distributed.run --nproc-per-node N` would.  Rank 0 inherits stdout and prints the JSON line."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in pending:          # one rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    raise SystemExit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default=WORKLOADS[0], choices=WORKLOADS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=6144, help="primary rays in the cpu_baseline sample")
    ap.add_argument("--roofline-stride", type=int, default=16, help="every k-th pixel feeds the byte count")
    ap.add_argument("--no-scaling-proxy", action="store_true")
    ap.add_argument("--inflight", type=int, default=None, choices=[1, 2, 3, 4],
                    help="render calls in flight (rtx_render_*_enqueue): they run side by side on the GPU, each on its own streams; "
                         "default 2 for batches of frames, 4 for the passes of a frame with several samples per pixel (measured)")
    ap.add_argument("--batch", type=int, default=8,
                    help="frames per render call (rtx_render_tiles_batch_dev_enqueue, 1-16): one set of launches traces them all")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                      # does not return
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import torch
    from raytracing_project_amd import api, scenes, tiles

    # RTX_BENCH_REHEARSE=1: all ranks share GPU 0 and exchange over gloo through host memory -- a dry run
    # of the N > 1 control flow on a one-GPU box (RCCL refuses two ranks on one device).  Its line says
    # so ("rehearsal": true) and is not a measurement.
    rehearse = world > 1 and os.environ.get("RTX_BENCH_REHEARSE") == "1"
    dist = None
    if rehearse:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local_rank)
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    wl = build_workload(args.workload, api, scenes)
    W, H = wl["W"], wl["H"]
    scene = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
    scene.set_lights(wl["lights"])
    scene.commit(local_rank)
    info = scene.info()
    opts = api.Options(*wl["opts"])
    offsets = spp_offsets(wl["spp"])
    cams = [api.make_camera(wl["view"], W, H, subpixel=o) for o in offsets]
    rect = (0, 0, W, H)
    kmax = tiles.max_local_tiles(W, H, TILE, world)
    # one rank owns the whole frame: render it row-major in place, nothing to gather or reassemble
    npix_local = kmax * TILE * TILE if world > 1 else W * H
    pipelined = len(cams) == 1
    B = max(1, min(args.batch, api.MAX_BATCH)) if pipelined else 1
    if args.inflight is None:
        args.inflight = 2 if pipelined else 4
    # Three sets of output buffers: the render of frame k+1 (librtx's own stream) overlaps the gather
    # and reassembly of frame k (torch's streams); a set is reused only after the torch work that
    # read it has finished (event per set).
    NSETS = max(3, args.inflight + 2)
    sets = []
    for _ in range(NSETS):
        b = {"rgb": torch.zeros((B * npix_local, 3), dtype=torch.float32, device=dev),
             "face": torch.full((B * npix_local, 1), -1, dtype=torch.int32, device=dev),
             "free": torch.cuda.Event()}
        b["acc"] = torch.zeros_like(b["rgb"]) if len(cams) > 1 else None
        if world > 1 and rank == 0:
            b["img"] = torch.empty((B, H, W, 3), dtype=torch.float32, device=dev)
        if world > 1 and rank == 0 and not rehearse:
            b["g_rgb"] = torch.empty((world, B * npix_local, 3), dtype=torch.float32, device=dev)
        b["free"].record()
        sets.append(b)
    state = {"k": 0}

    def exchange(b, out, face, nf=1):
        """The finished local pixels of nf frames -> the last of those frames on rank 0 (None elsewhere)."""
        if world == 1:
            return (out.view(-1, H, W, 3)[nf - 1], face.view(-1, H, W, 1)[nf - 1])
        # several ranks: the frames that travel are the colour images (what raytraceScene produces); hit ids
        # stay where they were computed (12 instead of 16 bytes per pixel into rank 0 and through assemble).
        # One gather per render call: all nf frames of the batch at once.
        out = out[: nf * npix_local]
        if rehearse:
            g_rgb = tiles.gather_to_root(out.cpu(), world, rank, dist)
            if rank == 0:
                g_rgb = g_rgb.to(dev)
        else:
            g_rgb = tiles.gather_to_root(out, world, rank, dist, b["g_rgb"][:, : nf * npix_local] if rank == 0 else None)
        if rank == 0:
            # one pass per frame back to row-major (rtx_tiles_assemble_dev on torch's stream; tiles.assemble is its torch twin)
            stride = g_rgb.stride(0) // 3
            for f in range(nf):
                api.tiles_assemble_dev(g_rgb.data_ptr() + f * npix_local * 12, world, stride, rect, TILE, 3, b["img"][f].data_ptr(),
                                       torch.cuda.current_stream().cuda_stream)
            return (b["img"][nf - 1], None)
        return None

    spp_rgb = [torch.zeros((npix_local, 3), dtype=torch.float32, device=dev) for _ in cams] if len(cams) > 1 else None

    def frame():
        """One frame of several samples per pixel: the passes (one camera per sub-pixel offset) are enqueued side by side,
        each into a buffer of its own, then summed in pass order and averaged; gather to rank 0, assemble."""
        b = sets[state["k"] % NSETS]
        state["k"] += 1
        face, acc = b["face"], b["acc"]
        rays = [0, 0, 0]
        ms = [0.0, 0.0, 0.0]
        b["free"].synchronize()            # torch is done with this set (frame k - 2)
        torch.cuda.current_stream().synchronize()       # ... and with the pass buffers (the sum of the frame before)

        def finish_one():
            st = scene.render_finish()
            rays[0] += st.rays_primary
            rays[1] += st.rays_shadow
            rays[2] += st.rays_reflection
            ms[0] += st.ms_trace_camera
            ms[1] += st.ms_trace - st.ms_trace_camera
            ms[2] += st.ms_total
        flying = 0
        for k, cam in enumerate(cams):
            if world > 1:
                scene.render_tiles_dev_enqueue(cam, opts, rect, TILE, rank, world, spp_rgb[k].data_ptr(), None, face.data_ptr() if k == 0 else None)
            else:
                scene.render_dev_enqueue(cam, opts, rect, spp_rgb[k].data_ptr(), None, face.data_ptr() if k == 0 else None)
            flying += 1
            if flying >= args.inflight:
                finish_one()
                flying -= 1
        for _ in range(flying):
            finish_one()
        acc.zero_()
        for k in range(len(cams)):
            acc.add_(spp_rgb[k])
        out = acc.mul_(1.0 / len(cams))
        img = exchange(b, out, face)
        b["free"].record()
        return rays, ms, img

    # Frames of a sequence go to the library B at a time (one camera per frame -- the same one here, the library
    # does not know that): one set of launches traces and shades the rank's tiles of all B frames.
    host = {"wait_buffers": 0.0, "enqueue": 0.0, "wait_render": 0.0, "exchange": 0.0}    # where rank 0's host time goes (seconds)

    def batch_enqueue(nf):
        b = sets[state["k"] % NSETS]
        state["k"] += 1
        t_a = time.perf_counter()
        b["free"].synchronize()
        t_b = time.perf_counter()
        host["wait_buffers"] += t_b - t_a
        b["nf"] = nf
        scene.render_tiles_batch_dev([cams[0]] * nf, opts, rect, TILE if world > 1 else 0, rank if world > 1 else 0, world,
                                     b["rgb"].data_ptr(), None, b["face"].data_ptr(), enqueue=True)
        host["enqueue"] += time.perf_counter() - t_b
        return b

    def batch_finish(b):
        t_a = time.perf_counter()
        st = scene.render_finish()
        t_b = time.perf_counter()
        host["wait_render"] += t_b - t_a
        rays = [st.rays_primary, st.rays_shadow, st.rays_reflection]
        ms = [st.ms_trace_camera, st.ms_trace - st.ms_trace_camera, st.ms_total]
        img = exchange(b, b["rgb"], b["face"], b["nf"])
        b["free"].record()
        host["exchange"] += time.perf_counter() - t_b
        return rays, ms, img

    def frames(n):
        """n frames; returns (rays, ms, img) per render call (per frame, or per batch of up to B frames)."""
        out = []
        if not pipelined:
            return [frame() for _ in range(n)]
        flying = []
        left = n
        ncalls = (n + B - 1) // B                # calls of (nearly) equal size: 20 frames at B = 8 go as 7 + 7 + 6
        while left > 0:
            nf = (left + ncalls - 1) // ncalls
            ncalls -= 1
            flying.append(batch_enqueue(nf))
            left -= nf
            if len(flying) >= args.inflight:
                out.append(batch_finish(flying.pop(0)))
        while flying:
            out.append(batch_finish(flying.pop(0)))
        return out

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # setup, not steps: the library times its wave schedulers on its first seven render calls of a
    # scene and keeps the fastest per launch kind; let that settle before the warmup / timed steps
    frames(7 * B)
    frames(args.warmup)
    sync()
    for k in host:
        host[k] = 0.0
    t0 = time.perf_counter()
    rays_tot = np.zeros(3, np.int64)
    ms_tot = np.zeros(3, np.float64)
    img = None
    for r, m, img in frames(args.steps):
        rays_tot += r
        ms_tot += m
    t_loop = time.perf_counter() - t0
    sync()
    dt = time.perf_counter() - t0
    host_ms = {k: v / args.steps * 1e3 for k, v in host.items()}
    host_ms["final_sync"] = (dt - t_loop) / args.steps * 1e3
    # whole-job aggregates: rays summed over ranks, time = max over ranks
    agg = torch.tensor([float(rays_tot[0]), float(rays_tot[1]), float(rays_tot[2])], dtype=torch.float64, device="cpu" if rehearse else dev)
    tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
    if dist is not None:
        dist.all_reduce(agg)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_rays = float(agg.sum().item())
    dt = float(tmax.item())
    # Per-kernel durations for the roofline: frames in flight share the GPU, so the HIP-event spans of the timed
    # region include time given to the other frames' kernels.  The same frames once more, one at a time
    # (synchronous calls, nothing beside them): these spans are each kernel's own duration.
    # (the same launches as in the timed region: B frames per call)
    ms_iso = None
    if world == 1:
        ms_iso = np.zeros(3, np.float64)
        ncalls = max(2, args.steps // B)
        for _ in range(ncalls):
            for cam in cams:
                scene.render_tiles_batch_dev([cam] * B, opts, rect, 0, 0, 1, sets[0]["rgb"].data_ptr(), None, sets[0]["face"].data_ptr())
                st = scene.stats()
                ms_iso += [st.ms_trace_camera, st.ms_trace - st.ms_trace_camera, st.ms_total]
        ms_iso /= ncalls * B
    # pixels whose primary ray hit geometry, from the hit ids of the last frame on every rank
    last = sets[(state["k"] - 1) % NSETS]["face"][:npix_local]
    nhit = torch.tensor([float((last >= 0).sum().item())], dtype=torch.float64, device="cpu" if rehearse else dev)
    if dist is not None:
        dist.all_reduce(nhit)

    result = None
    if rank == 0:
        hits = int(nhit.item())
        result = {
            "metric": "Mrays/sec (primary+shadow+reflection)", "value": total_rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "triangles": int(info.ntris), "bvh_nodes": int(info.nnodes),
                       "bvh_build_ms_host": float(info.build_ms), "scene_device_bytes": int(info.device_bytes),
                       "triangles_outside_bvh": int(info.nwild),
                       "image": [W, H], "lights": len(wl["lights"]), "spp": wl["spp"], "max_bounces": wl["opts"][0],
                       "tile": TILE, "partition": (("image tiles round-robin over %d ranks, scene replicated, RGB gathered to rank 0" % world) if world > 1
                                     else "one rank renders the whole frame row-major") + (("; %d frames per render call (rtx_render_tiles_batch_dev_enqueue), %d calls in flight" % (B, args.inflight)) if pipelined
                                        else "; %d passes per frame (one per sub-pixel sample), %d in flight, summed in pass order" % (len(cams), args.inflight)),
                       "frames_per_call": B, "calls_in_flight": args.inflight,
                       "rays_per_frame": {"primary": int(agg[0].item() / args.steps), "shadow": int(agg[1].item() / args.steps),
                                          "reflection": int(agg[2].item() / args.steps)},
                       "pixels_hitting_geometry": hits,
                       # rank 0's host thread, ms per frame of the timed region: waiting for a free buffer set (torch still gathering /
                       # reassembling an earlier call), enqueueing, waiting for the render, issuing gather + reassembly, the last sync
                       "host_ms_per_frame_rank0": host_ms,
                       # spans of the timed region: with several frames in flight they overlap each other
                       "kernel_span_ms_per_frame_rank0": {"trace_camera": ms_tot[0] / args.steps,
                                                           "trace_queue": ms_tot[1] / args.steps,
                                                           "all_kernels": ms_tot[2] / args.steps}},
        }
    if rank == 0 and rehearse:
        result["config"]["rehearsal"] = True
        if len(cams) == 1:      # the reassembled parts against one full-frame render, bit for bit
            f_rgb = torch.zeros((W * H, 3), dtype=torch.float32, device=dev)
            f_face = torch.full((W * H, 1), -1, dtype=torch.int32, device=dev)
            scene.render_dev(cams[0], opts, rect, f_rgb.data_ptr(), None, f_face.data_ptr())
            result["config"]["rehearsal_equals_full_frame"] = bool(
                hits == int((f_face >= 0).sum().item()) and torch.equal(img[0].reshape(-1, 3).view(torch.int32), f_rgb.view(torch.int32)))
    if rank == 0 and world == 1 and not args.no_scaling_proxy:
        result["scaling_proxy"] = scaling_proxy(scene, cams[0], opts, rect, W, H, torch, dev, tiles, max(5, args.steps), args.inflight, B)
    if rank == 0 and world == 1 and not args.no_roofline:
        result["config"]["kernel_ms_per_frame_one_call_at_a_time"] = {"trace_camera": ms_iso[0], "trace_queue": ms_iso[1], "all_kernels": ms_iso[2]}
        result["roofline"] = roofline(args, wl, scene, info, ms_iso, rays_tot / args.steps, B)
        ent = result["roofline"]["rays_entering_bvh_per_frame"]
        # rays that pass the reference's root-box gate and walk the hierarchy (the rest end at the gate: the camera of
        # SURVEY 8(d) sees past the soup on more than half of the frame)
        result["config"]["rays_entering_bvh_per_frame"] = ent
        result["mrays_per_s_entering_bvh"] = (ent["primary"] + ent["shadow"] + result["config"]["rays_per_frame"]["reflection"]) / (dt / args.steps) / 1e6
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args, wl)
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    scene.close()


def scaling_proxy(scene, cam, opts, rect, W, H, torch, dev, tiles, steps, inflight=2, batch=1):
    """Strong scaling rehearsed on one GPU: the share of rank 0 of N (rtx_render_tiles_dev, part 0 of N, the
    call every rank of an N-GPU run makes) timed alone, in batches and with calls in flight like the bench loop.  With perfect
    scaling part 0 of N costs 1/N of part 0 of 1; the ratio is the render-side bound on the N-GPU speed-up
    (the RGB gather and reassembly on rank 0 come on top and overlap the next frame's render)."""
    ms = {}
    for nparts in (1, 2, 4, 8):
        kmax = tiles.max_local_tiles(W, H, TILE, nparts)
        rgb = torch.zeros((batch * kmax * TILE * TILE, 3), dtype=torch.float32, device=dev)
        face = torch.full((batch * kmax * TILE * TILE, 1), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()

        def run(n):
            flying = 0
            left = n
            while left > 0:
                nf = min(batch, left)
                left -= nf
                scene.render_tiles_batch_dev([cam] * nf, opts, rect, TILE, 0, nparts, rgb.data_ptr(), None, face.data_ptr(), enqueue=True)
                flying += 1
                if flying >= inflight:
                    scene.render_finish()
                    flying -= 1
            for _ in range(flying):
                scene.render_finish()
        run(8 * batch)                           # the scheduler tuner starts over when the ray count changes 2x
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        ms[nparts] = (time.perf_counter() - t0) / steps * 1e3
    return {"ms_per_frame_part0_of_N": {str(n): ms[n] for n in ms},
            "render_speedup_bound": {str(n): ms[1] / ms[n] for n in ms},
            "note": "one GPU renders rank 0's tiles of an N-rank run (same tile size, same round-robin); "
                    "ms[1] / ms[N] bounds the N-GPU speed-up from the render side"}


def oracle_scene(wl):
    from oracle import oracle_api as oa
    oa.build()
    s = oa.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [oa.make_mtl(**m) for m in wl["mats"]])
    s.set_lights(wl["lights"])
    return oa, s


PMC_FILE = os.path.join("profiles", "r02_pmc.json")     # scripts/prof_derive.py --pmc-json over this round's rocprofv3 passes


def any_blocker_shadows(wl):
    """librtx stops a shadow ray at any blocker when every material shades a blocked light alike (rtx.hip, commit)."""
    m = wl["mats"]
    if not m:
        return False
    key = lambda d: tuple((k, tuple(np.ravel(d.get(k, dflt)).tolist())) for k, dflt in
                          (("ka", (.3, .3, .3)), ("kd", (.5, .5, .5)), ("ks", (1., 1., 1.)), ("ns", 10.0), ("d", 1.0)))
    tm = np.asarray(wl["tri_mat"])
    return all(key(x) == key(m[0]) for x in m) and bool((tm >= 0).all() and (tm < len(m)).all())


def roofline(args, wl, scene, info, ms, rays, frames_per_launch=1):
    """The dominant traversal kernel against the rooflines that could bound it.

    ms: per-frame kernel durations measured one render call at a time (HIP events on the library's stream); a call's
    launches carry frames_per_launch frames (rtx_render_tiles_batch_dev), and every per-launch figure below is for
    such a launch.
    * algorithmic_bytes_per_launch: SURVEY 8(d)'s figure, 32 B per node visited + 36 B per triangle tested + io,
      counted by the CPU mirror (oracle/rt_oracle_bvh.c) walking the same BVH in the same order on every k-th
      pixel's rays (shadow rays bounded at the light and stopped at any blocker exactly where the kernel does).
      It is LOAD DEMAND: nearly all of it is served by L1 / L2 / Infinity Cache, so demand / duration may exceed
      the HBM peak and is reported as `load_demand`, not as a fraction of HBM.
    * achieved / frac (bound "hbm"): the bytes that did cross the HBM-side interface per launch (`traffic`, PMC
      FETCH_SIZE x2 + WRITE_SIZE from this round's rocprofv3 passes of this command) / the launch duration
      measured here, against 8 TB/s; without a PMC record, the compulsory bytes (scene + io touched once).
    * valu: the bound that binds -- VALU-busy share of the kernel's SIMD cycles x active lanes per VALU
      instruction (same PMC passes): the fraction of the vector ALU's lane-cycles doing useful work.
    """
    oa, osc = oracle_scene(wl)
    W, H = wl["W"], wl["H"]
    stride = max(1, args.roofline_stride)
    nodes, order = scene.bvh()
    tris = scene.tris()
    lo, hi = osc.root_box()
    ocam = oa.make_camera(wl["view"], W, H)
    o, d1 = oa.camera_rays(ocam, (0, 0, W, H))
    pick = np.arange(0, W * H, stride)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(16, cores)
    t_cpu0 = time.perf_counter()
    t, f, nv, tt, gs, tf = oa.bvh_trace(nodes, tris, order, lo, hi, o[pick], d1[pick], threads=threads, nwild=info.nwild,
                                        layout_counts=True)
    cpu_bvh_s = time.perf_counter() - t_cpu0
    prim_bytes = (32.0 * nv.sum() + 36.0 * tt.sum()) * stride + 8.0 * W * H
    # what the device layout really loads: 64-B packed groups, 16 B for a triangle's plane stage, 48 B more when it
    # goes on to the barycentrics, 8 B result
    prim_layout = (64.0 * gs.sum() + 16.0 * tt.sum() + 48.0 * tf.sum()) * stride + 8.0 * W * H
    entering = {"primary": int((gs > 0).sum()) * stride, "shadow": 0}
    per_ray = {"primary_nodes": float(nv.mean()), "primary_tris": float(tt.mean()),
               "primary_nodes_of_rays_entering_bvh": float(nv[gs > 0].mean()) if (gs > 0).any() else 0.0}
    hit = f >= 0
    sh_bytes = sh_layout = 0.0
    anyhit = any_blocker_shadows(wl)
    if hit.any() and wl["opts"][4]:
        # hit points / shadow directions in numpy fp32: ulp-level differences from the kernel's own values do not
        # move the visit counts.  First light only, scaled by the light count.
        dd = d1[pick][hit]
        d2 = dd / np.sqrt((dd * dd).sum(1, dtype=np.float32))[:, None]
        P = (o[pick][hit] + t[hit, None] * d2).astype(np.float32)
        L = np.float32(wl["lights"][0])
        sd = (L[None] - P).astype(np.float32)
        dist = np.sqrt((sd * sd).sum(1, dtype=np.float32)).astype(np.float32)
        sd1 = (sd / dist[:, None]).astype(np.float32)
        tmax = (dist * np.float32(1.001)).astype(np.float32)             # shadow_tmax(), rtx_kernels.h
        _, _, nv2, tt2, gs2, tf2 = oa.bvh_trace_shadow(nodes, tris, order, lo, hi, P, sd1, tmax, anyhit, threads=threads, nwild=info.nwild)
        nl = len(wl["lights"])
        sh_bytes = ((32.0 * nv2.sum() + 36.0 * tt2.sum()) + 72.0 * len(P)) * stride * nl
        sh_layout = ((64.0 * gs2.sum() + 16.0 * tt2.sum() + 48.0 * tf2.sum()) + 72.0 * len(P)) * stride * nl
        entering["shadow"] = int((gs2 > 0).sum()) * stride * nl
        per_ray.update({"shadow_nodes": float(nv2.mean()), "shadow_tris": float(tt2.mean()),
                        "shadow_rays_stop_at_any_blocker": bool(anyhit)})
    spp = wl["spp"]
    B = frames_per_launch
    cand = [("k_trace_camera", prim_bytes * B, prim_layout * B, ms[0] / spp * B), ("k_trace_queue", sh_bytes * B, sh_layout * B, ms[1] / spp * B)]
    if wl["opts"][0] > 0:     # reflection launches share k_trace_queue's time; only the camera kernel is attributable
        cand = cand[:1]
    name, nbytes, lbytes, kms = max(cand, key=lambda c: c[3])
    scene_bytes = 64.0 * (info.nnodes / 4.0) + 64.0 * info.ntris + 4.0 * info.ntris
    io_bytes = (8.0 * W * H if name == "k_trace_camera" else 40.0 * rays[1] / spp) * B
    compulsory = scene_bytes + io_bytes
    pmc = {}
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as fh:
            pmc = json.load(fh).get(args.workload, {}).get(name, {})
    except (OSError, ValueError):
        pass
    traffic = pmc.get("hbm_bytes_per_launch")
    sec = kms * 1e-3
    hbm_bytes = traffic if traffic is not None else compulsory
    res = {"bound": "hbm", "kernel": name, "unit": "GB/s", "peak": HBM_PEAK_GBS,
           "achieved": hbm_bytes / sec / 1e9 if sec > 0 else 0.0,
           "traffic": traffic, "avg_launch_ms": kms, "frames_per_launch": B,
           "achieved_is": ("HBM-side bytes per launch (traffic: PMC FETCH_SIZE x2 + WRITE_SIZE, %s) / the launch duration measured in this run"
                           % PMC_FILE) if traffic is not None else "compulsory bytes per launch (no PMC record for this workload) / measured launch duration",
           "compulsory_bytes_per_launch": compulsory,
           "algorithmic_bytes_per_launch": nbytes,
           "bytes_model": "SURVEY 8(d): 32 B/node visited + 36 B/triangle tested + io (8 B/primary, 72 B/queued ray); counts from the CPU "
                          "mirror traversal of the same BVH on every %d-th pixel's rays, scaled" % stride,
           "per_ray": per_ray}
    res["frac"] = res["achieved"] / HBM_PEAK_GBS
    # the 8(d) bytes are load demand served by the caches: set against the L2's aggregate bandwidth, never against HBM
    L2_PEAK_GBS = 34500.0           # MI355X_MICROARCH.md: 8 x 4 MiB L2, ~34.5 TB/s aggregate
    res["load_demand"] = {"GB/s": nbytes / sec / 1e9 if sec > 0 else 0.0, "device_layout_GB/s": lbytes / sec / 1e9 if sec > 0 else 0.0,
                          "device_layout_bytes_per_launch": lbytes,
                          "frac_of_l2_peak": (lbytes / sec / 1e9 / L2_PEAK_GBS) if sec > 0 else 0.0, "l2_peak_GB/s": L2_PEAK_GBS,
                          "note": "served by L1 / L2 / Infinity Cache (scene resident: see traffic); not an HBM figure"}
    if traffic is not None:
        res["traffic_over_compulsory"] = traffic / compulsory
    if "valu_busy" in pmc and "active_lanes" in pmc:
        res["valu"] = {"bound": "valu", "busy": pmc["valu_busy"], "active_lanes": pmc["active_lanes"],
                       "frac": pmc["valu_busy"] * pmc["active_lanes"],
                       "waves_per_simd": pmc.get("waves_per_simd"), "ta_busy": pmc.get("ta_busy"), "td_busy": pmc.get("td_busy"),
                       "note": "the bound that binds: share of the kernel's SIMD cycles issuing vector ALU instructions x active lanes "
                               "per instruction (%s, rocprofv3 --pmc SQ passes of this command with --inflight 1: one call at a time)" % PMC_FILE}
        res["binding_bound"] = "valu"
    res["rays_entering_bvh_per_frame"] = {k: v * spp for k, v in entering.items()}
    # the same traversal of the same BVH on the host's cores (the "same algorithm on the CPU" figure of SURVEY 8(d))
    res["cpu_bvh_traversal"] = {"value": len(pick) / cpu_bvh_s / 1e6, "unit": "Mrays/s (primary)", "cores": threads, "host_cores": cores,
                                "sample": "every %d-th pixel's primary ray, oracle/rt_oracle_bvh.c" % stride}
    return res


def cpu_baseline(args, wl):
    """The reference CPU path (brute force over all triangles, oracle port) on a bounded sample."""
    oa, osc = oracle_scene(wl)
    W, H = wl["W"], wl["H"]
    ntri = len(wl["idx"])
    # ~15 s of work at ~6e7 triangle tests / s / core
    host_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(16, host_cores)
    budget = 6e7 * cores * 12.0
    n = int(max(256, min(args.cpu_sample * 64, budget / max(ntri, 1) / 1.6)))
    n = min(n, W * H)
    stride = max(1, (W * H) // n)
    pick = np.arange(0, W * H, stride)[:n]
    ocam = oa.make_camera(wl["view"], W, H)
    org, dirs = oa.camera_rays(ocam, (0, 0, W, H))
    org, dirs = org[pick].copy(), dirs[pick].copy()
    t0 = time.perf_counter()
    h = osc.closest(org, dirs, threads=cores)                    # primary rays, Flyscene::getIntersections
    hit = h["t"] != np.inf
    nrays = len(pick)
    if hit.any() and wl["opts"][4]:
        P = h["p"][hit]
        for L in wl["lights"][:1]:
            sd = (np.float32(L)[None] - P).astype(np.float32)
            osc.closest(P, sd, threads=cores)                    # shadow rays: closest hit, as canSeeLight does
            nrays += len(P)
    dt = time.perf_counter() - t0
    return {"value": nrays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "host_cores_available": host_cores, "kind": "port",
            "sample": "%d primary rays (every %d-th pixel) + %d shadow rays of %s, brute force over %d triangles "
                      "(reference algorithm, oracle/rt_oracle.c), %.1f s" % (len(pick), stride, nrays - len(pick),
                                                                             args.workload, ntri, dt)}


if __name__ == "__main__":
    main()

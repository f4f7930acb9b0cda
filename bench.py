#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + shadow + reflection) of the HIP ray-scene
intersection path, one process per GPU.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one frame of the workload, each frame from the next camera of a slow orbit around the scene (an
animation, not one view rendered again and again): every rank renders its image tiles
(rtx_render_tiles_batch_dev, scene replicated per GPU), then the frame is gathered to rank 0 (RCCL) as the
3-byte pixels the reference writes to result.ppm (rtx_rgb_to_ppm8_dev; --out-format f32 for the float frame)
and put back into row-major order.  The K timed steps (barrier + synchronise on both sides) are repeated
--repeats times; value / ms_per_step are the MEDIAN repeat's (every repeat is listed), value = rays traced by all
ranks in its K frames / wall time (max over ranks).  Total work is fixed as N grows: "strong" scaling.

The JSON line also carries
  single_frame  one frame per call, one call at a time: the latency figure Flyscene::raytraceScene corresponds to;
  verify        one batch re-rendered with RTX_FLAG_VERIFY: sampled rays re-checked by brute force over every triangle;
  roofline      for the dominant traversal kernel: HBM-side traffic / its HIP-event duration against 8 TB/s, the
                SURVEY 8(d) byte model beside it (survey_8d_frac: load demand, served by the caches), the VALU bound;
  cpu_baseline  the CPU oracle (brute-force reference path) timed on this host on a bounded
                sample of the same rays; roofline.cpu_bvh_full_frame = the same BVH walked by the host's cores.
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
                    view=scenes.look_at_view((0, 0, 3), (0, 0, 0)), path=((0, 0, 3), (0, 0, 0)), W=3840, H=2160, opts=(0, 1, 0, 0, 1, 1), spp=1)
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
                    view=scenes.look_at_view(c + np.array([0, 0, 2 * r]), c), path=(c + np.array([0, 0, 2 * r]), c), W=1920, H=1080,
                    opts=(0, 1, 0, 0, 1, 1), spp=1)
    if name == "hall262k_1080p_reflect3":
        verts, idx, tm = scenes.hall()
        wall = mk(ka=(.1, .1, .1), kd=(.6, .55, .5), ks=(.3, .3, .3), ns=20.0, ni=1.0, d=0.5, illum=2)
        col = mk(ka=(.1, .1, .1), kd=(.5, .6, .7), ks=(.6, .6, .6), ns=40.0, ni=1.0, d=0.5, illum=2)
        return dict(verts=verts, idx=idx, tri_mat=tm, mats=[wall, col], lights=[[0.0, 2.7, 0.0]],
                    view=scenes.look_at_view((0.3, 1.6, 7.0), (0.0, 1.2, 0.0)), path=((0.3, 1.6, 7.0), (0.0, 1.2, 0.0), 0.02), W=1920, H=1080,
                    opts=(3, 1, 0, 0, 1, 1), spp=1)
    if name == "planeobj_512_primary_shadow":
        z = np.load(os.path.join(ROOT, "tests", "golden", "scene_planeobj.npz"))
        mats = [mk(ka=tuple(m[0:3]), kd=tuple(m[3:6]), ks=tuple(m[6:9]), ns=float(m[9]), ni=float(m[10]), d=float(m[11]),
                   illum=int(m[12])) for m in z["mats"]]
        return dict(verts=z["verts"], idx=z["idx"], tri_mat=z["tri_mat"], mats=mats, lights=[[-1.0, 1.0, 1.0]],
                    view=scenes.default_view(), path=((0, 0, 2), (0, 0, 0)), W=512, H=512, opts=(0, 1, 0, 0, 1, 1), spp=1)
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


def frame_view(wl, scenes, k, static=False):
    """View matrix of frame k of the run: the workload's camera on a slow orbit about the scene (0.25 degrees per
    frame; the hall's camera, which stands inside, 0.02), or the one fixed view of SURVEY 8(d) with --static-camera."""
    if static or "path" not in wl or k == 0:
        return wl["view"]
    eye, target = wl["path"][0], wl["path"][1]
    return scenes.orbit_view(eye, target, k, wl["path"][2] if len(wl["path"]) > 2 else 0.25)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default=WORKLOADS[0], choices=WORKLOADS)
    ap.add_argument("--repeats", type=int, default=5, help="the K timed steps are run this many times (each bracketed by barrier + synchronise); "
                                                           "the median repeat is reported, all are listed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=6144, help="primary rays in the cpu_baseline sample")
    ap.add_argument("--roofline-stride", type=int, default=16, help="every k-th pixel feeds the byte count")
    ap.add_argument("--no-scaling-proxy", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the single-frame and verification passes")
    ap.add_argument("--fast-triangles", action="store_true", help="RTX_FLAG_FAST_TRIANGLES: the watertight Moeller-Trumbore functor instead of the reference's "
                                                                  "arithmetic (hit ids no longer bit-exact; a measurement mode, never the headline)")
    ap.add_argument("--device-build", action="store_true", help="RTX_FLAG_DEVICE_BUILD: the hierarchy built on the GPU (Morton-ordered LBVH) instead of the host's "
                                                                "binned-SAH build; same hits, a measurement of the other tree")
    ap.add_argument("--static-camera", action="store_true", help="every frame from the same camera (rounds 1-2 measured that)")
    ap.add_argument("--out-format", default="rgb8", choices=["rgb8", "f32"],
                    help="what the ranks of an N > 1 run hand to rank 0: the P3 bytes of result.ppm (3 B/pixel) or the float frame (12)")
    ap.add_argument("--inflight", type=int, default=None, choices=[1, 2, 3, 4],
                    help="render calls in flight (rtx_render_*_enqueue): they run side by side on the GPU, each on its own streams; "
                         "default 2 for batches of frames, 4 for the passes of a frame with several samples per pixel (measured)")
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per render call (rtx_render_tiles_batch_dev_enqueue, 1-64): one set of launches traces them all; "
                         "default 16 x ranks, at most 64 (a rank's share of 16 N frames is as many rays as 16 whole frames)")
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
    base_flags = (api.FLAG_FAST_TRIANGLES if args.fast_triangles else 0) | (api.FLAG_DEVICE_BUILD if args.device_build else 0)
    if base_flags:
        scene.set_flags(base_flags)
    scene.commit(local_rank)
    info = scene.info()
    opts = api.Options(*wl["opts"])
    offsets = spp_offsets(wl["spp"])
    rect = (0, 0, W, H)
    kmax = tiles.max_local_tiles(W, H, TILE, world)
    # one rank owns the whole frame: render it row-major in place, nothing to gather or reassemble
    npix_local = kmax * TILE * TILE if world > 1 else W * H
    pipelined = len(offsets) == 1
    if args.inflight is None:
        args.inflight = 2 if pipelined else 4
    if args.batch is None:
        args.batch = 16 * world
    B = max(1, min(args.batch, api.MAX_BATCH)) if pipelined else 1
    # at least `inflight` calls per timed repeat, so that calls do run side by side when K is small against the batch
    B = max(1, min(B, (args.steps + args.inflight - 1) // args.inflight))
    rgb8 = world > 1 and args.out_format == "rgb8"
    out_bpp = 3 if rgb8 else 12

    cam_cache = {}

    def cams_of(k):
        """The cameras of frame k: one per sub-pixel sample.  Made once per frame index: the camera path is input, like the
        scene, and is laid out before the timed region (numpy takes ~0.15 ms per rtx_camera record, which at 8 ranks is as
        long as the rank's share of a frame takes to render)."""
        c = cam_cache.get(k)
        if c is None:
            view = frame_view(wl, scenes, k, args.static_camera)
            c = cam_cache[k] = [api.make_camera(view, W, H, subpixel=o) for o in offsets]
        return c

    # Output buffers in sets: the render of call k+1 (librtx's own streams) overlaps the gather and reassembly of call k
    # (torch's streams); a set is reused only after the torch work that read it has finished (event per set).
    NSETS = max(3, args.inflight + 2)
    sets = []
    for _ in range(NSETS):
        b = {"rgb": torch.zeros((B * npix_local, 3), dtype=torch.float32, device=dev),
             "face": torch.full((B * npix_local, 1), -1, dtype=torch.int32, device=dev),
             "free": torch.cuda.Event()}
        b["acc"] = torch.zeros_like(b["rgb"]) if not pipelined else None
        if rgb8:
            b["rgb8"] = torch.zeros((B * npix_local, 3), dtype=torch.uint8, device=dev)
        if world > 1 and rank == 0:
            b["img"] = torch.empty((B, H, W, 3), dtype=torch.uint8 if rgb8 else torch.float32, device=dev)
        if world > 1 and rank == 0 and not rehearse:
            b["g"] = torch.empty((world, B * npix_local, 3), dtype=torch.uint8 if rgb8 else torch.float32, device=dev)
        b["free"].record()
        sets.append(b)
    state = {"k": 0}

    def exchange(b, out, face, nf=1):
        """The finished local pixels of nf frames -> the last of those frames on rank 0 (None elsewhere)."""
        if world == 1:
            return (out.view(-1, H, W, 3)[nf - 1], face.view(-1, H, W, 1)[nf - 1])
        # several ranks: what travels is the colour image (what raytraceScene produces) -- by default as the bytes the
        # reference writes to result.ppm, quantised where the pixels lie; hit ids stay where they were computed.
        # One gather per render call: all nf frames of the batch at once.
        out = out[: nf * npix_local]
        if rgb8:
            api.rgb_to_ppm8_dev(out.data_ptr(), nf * npix_local, b["rgb8"].data_ptr(), torch.cuda.current_stream().cuda_stream)
            out = b["rgb8"][: nf * npix_local]
        if rehearse:
            g = tiles.gather_to_root(out.cpu(), world, rank, dist)
            if rank == 0:
                g = g.to(dev)
        else:
            g = tiles.gather_to_root(out, world, rank, dist, b["g"][:, : nf * npix_local] if rank == 0 else None)
        if rank == 0:
            # one pass per frame back to row-major (rtx_tiles_assemble_*_dev on torch's stream; tiles.assemble is its torch twin)
            stride = g.stride(0) // 3
            for f in range(nf):
                api.tiles_assemble_bytes_dev(g.data_ptr() + f * npix_local * out_bpp, world, stride, rect, TILE, out_bpp, b["img"][f].data_ptr(),
                                             torch.cuda.current_stream().cuda_stream)
            return (b["img"][nf - 1], None)
        return None

    spp_rgb = [torch.zeros((npix_local, 3), dtype=torch.float32, device=dev) for _ in offsets] if not pipelined else None

    def frame(kf):
        """Frame kf of the path at several samples per pixel: the passes (one camera per sub-pixel offset) are enqueued side by
        side, each into a buffer of its own, then summed in pass order and averaged; gather to rank 0, assemble."""
        b = sets[state["k"] % NSETS]
        state["k"] += 1
        cams = cams_of(kf)
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

    # Frames of a sequence go to the library B at a time, one camera per frame: one set of launches traces and shades
    # the rank's tiles of all B frames.
    host = {"wait_buffers": 0.0, "enqueue": 0.0, "wait_render": 0.0, "exchange": 0.0}    # where rank 0's host time goes (seconds)

    def batch_enqueue(k0, nf):
        b = sets[state["k"] % NSETS]
        state["k"] += 1
        cams = [cams_of(k0 + f)[0] for f in range(nf)]
        t_a = time.perf_counter()
        b["free"].synchronize()
        t_b = time.perf_counter()
        host["wait_buffers"] += t_b - t_a
        b["nf"] = nf
        scene.render_tiles_batch_dev(cams, opts, rect, TILE if world > 1 else 0, rank if world > 1 else 0, world,
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

    def frames(n, k0=0):
        """Frames k0 .. k0 + n - 1 of the camera path; returns (rays, ms, img) per render call (per frame, or per batch of up to B frames)."""
        out = []
        if not pipelined:
            return [frame(k0 + i) for i in range(n)]
        flying = []
        left = n
        ncalls = (n + B - 1) // B                # calls of (nearly) equal size: 20 frames at B = 8 go as 7 + 7 + 6
        while left > 0:
            nf = (left + ncalls - 1) // ncalls
            ncalls -= 1
            flying.append(batch_enqueue(k0 + n - left, nf))
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
    # Every timed repeat renders the SAME K frames of the path (frames 0 .. K-1; frame 0 is SURVEY 8(d)'s camera), so that the
    # repeats differ by noise only; within a repeat every frame has its own camera.
    frames(7 * B)
    frames(args.warmup, max(0, args.steps - args.warmup))
    for k in range(args.steps):
        cams_of(k)                               # (the timed frames' camera records exist before the clock starts)
    R = max(1, args.repeats)
    rep_dt, rep_rays, rep_ms, rep_host = [], [], [], []
    img = None
    for _ in range(R):
        sync()
        for k in host:
            host[k] = 0.0
        t0 = time.perf_counter()
        rays_tot = np.zeros(3, np.int64)
        ms_tot = np.zeros(3, np.float64)
        for r, m, img in frames(args.steps):
            rays_tot += r
            ms_tot += m
        t_loop = time.perf_counter() - t0
        sync()
        dt = time.perf_counter() - t0
        hm = {k: v / args.steps * 1e3 for k, v in host.items()}
        hm["final_sync"] = (dt - t_loop) / args.steps * 1e3
        rep_dt.append(dt); rep_rays.append(rays_tot); rep_ms.append(ms_tot); rep_host.append(hm)
    # whole-job aggregates per repeat: rays summed over ranks, time = max over ranks
    cdev = "cpu" if rehearse else dev
    agg = torch.tensor(np.asarray(rep_rays, np.float64), dtype=torch.float64, device=cdev)          # [R, 3]
    tmax = torch.tensor(rep_dt, dtype=torch.float64, device=cdev)
    if dist is not None:
        dist.all_reduce(agg)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    agg = agg.cpu().numpy()
    tmax = tmax.cpu().numpy()
    values = [float(agg[i].sum() / tmax[i] / 1e6) for i in range(R)]
    med = int(np.argsort(values)[(R - 1) // 2])    # the repeat with the median rate (the lower one of an even count): frames of a camera path differ in rays
    dt = float(tmax[med])
    total_rays = float(agg[med].sum())
    rays_tot, ms_tot, host_ms = rep_rays[med], rep_ms[med], rep_host[med]

    # Per-kernel durations for the roofline: frames in flight share the GPU, so the HIP-event spans of the timed
    # region include time given to the other frames' kernels.  The same kind of calls once more, one at a time
    # (synchronous, nothing beside them): these spans are each kernel's own duration.
    ms_iso = None
    iso_views = []
    single = None
    verify = None
    if world == 1:
        ms_iso = np.zeros(3, np.float64)
        ncalls = max(2, args.steps // B)
        for c in range(ncalls):
            k0 = (c * B) % max(1, args.steps)
            for si in range(len(offsets)):
                cams = [cams_of(k0 + f)[si] for f in range(B)]
                scene.render_tiles_batch_dev(cams, opts, rect, 0, 0, 1, sets[0]["rgb"].data_ptr(), None, sets[0]["face"].data_ptr())
                st = scene.stats()
                ms_iso += [st.ms_trace_camera, st.ms_trace - st.ms_trace_camera, st.ms_total]
            if c == 0:
                iso_views = [frame_view(wl, scenes, k0 + f, args.static_camera) for f in range(B)]
        ms_iso /= ncalls * B
    if world == 1 and pipelined and not args.no_extras:
        # One frame per call, one call at a time: what Flyscene::raytraceScene (src/flyscene.cpp:395-439) corresponds to.
        def one(n):
            rays = 0
            for kf in range(n):
                scene.render_tiles_batch_dev(cams_of(kf), opts, rect, 0, 0, 1, sets[0]["rgb"].data_ptr(), None, sets[0]["face"].data_ptr())
                st = scene.stats()
                rays += st.rays_primary + st.rays_shadow + st.rays_reflection
            return rays
        one(10)                                   # the tuner's trials for this size class
        times = []
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rays1 = one(max(8, min(args.steps, 32)))
            torch.cuda.synchronize()
            times.append(((time.perf_counter() - t1) / max(8, min(args.steps, 32)), rays1 / max(8, min(args.steps, 32))))
        times.sort()
        single = {"ms_per_frame": times[1][0] * 1e3, "mrays_per_s": times[1][1] / times[1][0] / 1e6, "ms_per_frame_min": times[0][0] * 1e3,
                  "note": "one frame per render call, one call at a time (host wait included): the latency of a single raytraceScene"}
        # The timed configuration once more with the standing check on: every camera launch re-checks ~2,000 of its rays by
        # brute force over every triangle (the reference's own algorithm); a mismatch would be a hit the hierarchy lost.
        scene.set_flags(api.FLAG_VERIFY | base_flags)
        cams = [cams_of(f)[0] for f in range(B)]
        scene.render_tiles_batch_dev(cams, opts, rect, 0, 0, 1, sets[0]["rgb"].data_ptr(), None, sets[0]["face"].data_ptr())
        st = scene.stats()
        scene.set_flags(base_flags)
        verify = {"checked": int(st.verify_checked), "mismatches": int(st.verify_mismatches), "rays_out_of_domain": int(st.rays_out_of_domain),
                  "note": "RTX_FLAG_VERIFY on one batch of %d frames: sampled camera rays re-traced by brute force over all triangles" % B}
    # pixels whose primary ray hit geometry, from the hit ids of the last frame on every rank
    lb = sets[(state["k"] - 1) % NSETS]
    nf_last = lb.get("nf", 1) if pipelined and world > 1 else 1          # (a lone rank's later passes reuse set 0 for whole batches: frame 0)
    last = lb["face"][(nf_last - 1) * npix_local: nf_last * npix_local]
    nhit = torch.tensor([float((last >= 0).sum().item())], dtype=torch.float64, device=cdev)
    if dist is not None:
        dist.all_reduce(nhit)

    result = None
    if rank == 0:
        hits = int(nhit.item())
        gather_bytes = 0 if world == 1 else int((world - 1) * npix_local * out_bpp)
        result = {
            "metric": "Mrays/sec (primary+shadow+reflection)", "value": total_rays / dt / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "repeats": {"n": R, "reported": "median", "value": sorted(values), "ms_per_step": sorted(float(x) / args.steps * 1e3 for x in tmax),
                        "spread": (max(values) - min(values)) / float(np.median(values)) if R > 1 else 0.0},
            "config": {"workload": args.workload, "triangles": int(info.ntris), "bvh_nodes": int(info.nnodes),
                       "bvh_builder": "device: Morton-ordered radix tree, expected-cost collapse (RTX_FLAG_DEVICE_BUILD)" if info.build_kind else "host: binned SAH, 16 threads",
                       "bvh_build_ms_host": None if info.build_kind else float(info.build_ms), "bvh_build_ms_device": float(info.device_build_ms) if info.build_kind else None,
                       "bvh_group_levels": int(info.max_depth), "scene_device_bytes": int(info.device_bytes),
                       "triangles_outside_bvh": int(info.nwild),
                       "image": [W, H], "lights": len(wl["lights"]), "spp": wl["spp"], "max_bounces": wl["opts"][0],
                       "camera": "one fixed view" if args.static_camera else "a camera path: every frame from the next position of an orbit about the scene (%.2f degrees per frame)"
                                 % (wl["path"][2] if len(wl.get("path", ())) > 2 else 0.25),
                       "triangle_test": "watertight Moeller-Trumbore (RTX_FLAG_FAST_TRIANGLES): NOT the reference's arithmetic" if args.fast_triangles
                                        else "the reference's plane + barycentric arithmetic (bit-exact hit ids)",
                       "tile": TILE, "partition": (("image tiles round-robin over %d ranks, scene replicated, %s gathered to rank 0" % (world, "P3 bytes (rtx_rgb_to_ppm8_dev)" if rgb8 else "float RGB")) if world > 1
                                     else "one rank renders the whole frame row-major") + (("; %d frames per render call (rtx_render_tiles_batch_dev_enqueue), %d calls in flight" % (B, args.inflight)) if pipelined
                                        else "; %d passes per frame (one per sub-pixel sample), %d in flight, summed in pass order" % (len(offsets), args.inflight)),
                       "frames_per_call": B, "calls_in_flight": args.inflight,
                       "gather": {"format": ("rgb8" if rgb8 else "f32") if world > 1 else None, "bytes_per_pixel": out_bpp if world > 1 else None,
                                  "bytes_into_rank0_per_frame": gather_bytes},
                       "rays_per_frame": {"primary": int(agg[med][0] / args.steps), "shadow": int(agg[med][1] / args.steps),
                                          "reflection": int(agg[med][2] / args.steps)},
                       "pixels_hitting_geometry": hits,
                       # rank 0's host thread, ms per frame of the timed region: waiting for a free buffer set (torch still gathering /
                       # reassembling an earlier call), enqueueing, waiting for the render, issuing gather + reassembly, the last sync
                       "host_ms_per_frame_rank0": host_ms,
                       # spans of the timed region: with several frames in flight they overlap each other
                       "kernel_span_ms_per_frame_rank0": {"trace_camera": ms_tot[0] / args.steps,
                                                           "trace_queue": ms_tot[1] / args.steps,
                                                           "all_kernels": ms_tot[2] / args.steps}},
        }
        if single is not None:
            result["single_frame"] = single
        if verify is not None:
            result["verify"] = verify
    if rank == 0 and world == 1 and pipelined and not args.no_extras:
        try:        # (an extra: whatever happens in it, the line above is what was measured)
            result["device_build"] = device_build_figures(api, torch, wl, scene, info, opts, rect, lambda k: cams_of(k)[0], sets[0])
        except Exception as e:  # noqa: BLE001
            result["device_build"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and rehearse:
        result["config"]["rehearsal"] = True
        if pipelined:      # the reassembled parts against one full-frame render of the same camera, bit for bit
            f_rgb = torch.zeros((W * H, 3), dtype=torch.float32, device=dev)
            f_face = torch.full((W * H, 1), -1, dtype=torch.int32, device=dev)
            scene.render_dev(cams_of(args.steps - 1)[0], opts, rect, f_rgb.data_ptr(), None, f_face.data_ptr())
            want = f_rgb
            if rgb8:
                want = torch.zeros((W * H, 3), dtype=torch.uint8, device=dev)
                api.rgb_to_ppm8_dev(f_rgb.data_ptr(), W * H, want.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            got = img[0].reshape(-1, 3)
            same = torch.equal(got, want) if rgb8 else torch.equal(got.view(torch.int32), want.view(torch.int32))
            result["config"]["rehearsal_equals_full_frame"] = bool(hits == int((f_face >= 0).sum().item()) and same)
    if rank == 0 and world == 1 and not args.no_scaling_proxy and pipelined:
        result["scaling_proxy"] = scaling_proxy(scene, lambda k: cams_of(k)[0], opts, rect, W, H, torch, dev, tiles, args.inflight,
                                                max(1, args.batch // world))
    if rank == 0 and world == 1 and not args.no_roofline:
        result["config"]["kernel_ms_per_frame_one_call_at_a_time"] = {"trace_camera": ms_iso[0], "trace_queue": ms_iso[1], "all_kernels": ms_iso[2]}
        result["roofline"] = roofline(args, wl, scene, info, ms_iso, rays_tot / args.steps, B, iso_views)
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


def device_build_figures(api, torch, wl, scene, info, opts, rect, cam_of, bufs):
    """The hierarchy built / refitted on the GPU (RTX_FLAG_DEVICE_BUILD, rtx_scene_update_vertices_dev) on this workload's
    mesh: wall times with the vertices already on the device, and an animated frame -- new vertices, refit, one render
    call -- on a second handle of the scene (the timed scene above is left alone)."""
    d = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
    d.set_lights(wl["lights"])
    d.set_flags(api.FLAG_DEVICE_BUILD)
    d.commit(torch.cuda.current_device())
    di = d.info()
    v = torch.from_numpy(np.ascontiguousarray(wl["verts"], np.float32)).cuda()
    amp = 1e-3 * float(np.abs(wl["verts"]).max())

    def moved(k):          # the mesh of frame k: every vertex displaced a little along a travelling wave
        return v + amp * torch.sin(v[:, [1, 2, 0]] * 3.0 + 0.37 * k)
    def timed(how, n):
        ts = []
        for k in range(n):
            nv = moved(k)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); d.update_vertices(nv, how); ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts[1:]))
    rebuild = timed(api.UPDATE_REBUILD, 6)
    refit = timed(api.UPDATE_REFIT, 8)
    for k in range(10):   # the tuner's trials on this handle
        d.render_tiles_batch_dev([cam_of(k)], opts, rect, 0, 0, 1, bufs["rgb"].data_ptr(), None, bufs["face"].data_ptr())
    n = 24
    rays = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        nv = moved(k)
        torch.cuda.synchronize()        # (the library's streams do not wait for torch's: the vertices have to be there, as for every _dev entry point)
        d.update_vertices(nv, api.UPDATE_REFIT)
        d.render_tiles_batch_dev([cam_of(k)], opts, rect, 0, 0, 1, bufs["rgb"].data_ptr(), None, bufs["face"].data_ptr())
        st = d.stats()
        rays += st.rays_primary + st.rays_shadow + st.rays_reflection
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"build_ms_first": float(di.device_build_ms), "rebuild_ms": rebuild, "refit_ms": refit, "host_build_ms": float(info.build_ms) if not info.build_kind else None,
           "groups": int(di.nnodes // 4), "group_levels": int(di.max_depth), "built_on_device": bool(di.build_kind),
           "animated_frame": {"ms_per_frame": dt / n * 1e3, "mrays_per_s": rays / dt / 1e6,
                              "refits": int(d.info().refits),
                              "note": "every frame: new vertex positions (a device tensor computed by torch), rtx_scene_update_vertices_dev(REFIT), one render call, one call at a time"},
           "note": "Morton-ordered radix tree + expected-cost collapse built on the GPU, wall ms with the vertices resident (median; the first build also loads the sort's kernels); "
                   "the timed scene above uses the %s tree" % ("device-built" if info.build_kind else "host's binned-SAH")}
    d.close()
    return out


def scaling_proxy(scene, cam_of, opts, rect, W, H, torch, dev, tiles, inflight=2, per_rank=16):
    """Strong scaling rehearsed on one GPU: the share of rank 0 of N (rtx_render_tiles_batch_dev, part 0 of N, the
    call every rank of an N-GPU run makes) timed alone, the way the bench loop of an N-rank run issues it: 16 N frames
    per call, at most 64 (a share of 16 N frames holds as many rays as 16 whole frames), `inflight` calls in flight, every frame from
    the next camera of the path.  With perfect scaling part 0 of N costs 1/N of part 0 of 1; the ratio is the render-side
    bound on the N-GPU speed-up (the gather and reassembly on rank 0 come on top and overlap the next call's render).
    Each N is timed three times over >= 128 frames (the spans are short: one repeat of 20 frames of a 1/8 share is 4 ms
    and its timing moved by 15 % from run to run); min and median are printed."""
    from raytracing_project_amd import api
    ms, ms_med, spans, batch_of, gather = {}, {}, {}, {}, {}
    k0 = 100000                                   # (camera indices of their own: these frames belong to no timed step)
    for nparts in (1, 2, 4, 8):
        batch = min(api.MAX_BATCH, per_rank * nparts)
        kmax = tiles.max_local_tiles(W, H, TILE, nparts)
        npl = kmax * TILE * TILE if nparts > 1 else W * H
        rgb = torch.zeros((batch * npl, 3), dtype=torch.float32, device=dev)
        face = torch.full((batch * npl, 1), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        acc = {"cam": 0.0, "queue": 0.0, "all": 0.0, "frames": 0}

        def run(n, k, path=None):
            flying = []
            left = n
            while left > 0:
                nf = min(batch, left)
                left -= nf
                cams = path[n - left - nf: n - left] if path is not None else [cam_of(k + f) for f in range(nf)]
                k += nf
                scene.render_tiles_batch_dev(cams, opts, rect, TILE if nparts > 1 else 0, 0, nparts, rgb.data_ptr(), None, face.data_ptr(), enqueue=True)
                flying.append(nf)
                if len(flying) >= inflight:
                    st = scene.render_finish(); nfd = flying.pop(0)
                    acc["cam"] += st.ms_trace_camera; acc["queue"] += st.ms_trace - st.ms_trace_camera; acc["all"] += st.ms_total; acc["frames"] += nfd
            while flying:
                st = scene.render_finish(); nfd = flying.pop(0)
                acc["cam"] += st.ms_trace_camera; acc["queue"] += st.ms_trace - st.ms_trace_camera; acc["all"] += st.ms_total; acc["frames"] += nfd
        run(8 * batch, k0)                        # the scheduler tuner's trials for this size class
        n = max(128, 2 * inflight * batch)
        n = (n + batch - 1) // batch * batch
        t = []
        for rep in range(3):
            path = [cam_of(k0 + 1000 * (rep + 1) + f) for f in range(n)]      # (input, made before the clock starts, as in the timed steps)
            torch.cuda.synchronize()
            acc.update(cam=0.0, queue=0.0, all=0.0, frames=0)
            t0 = time.perf_counter()
            run(n, k0 + 1000 * (rep + 1), path)
            torch.cuda.synchronize()
            t.append((time.perf_counter() - t0) / n * 1e3)
        t.sort()
        ms[nparts], ms_med[nparts] = t[0], t[1]
        spans[nparts] = {"trace_camera": acc["cam"] / acc["frames"], "trace_queue": acc["queue"] / acc["frames"], "all_kernels": acc["all"] / acc["frames"]}
        batch_of[nparts] = batch
        gather[nparts] = {"rgb8": int((nparts - 1) * npl * 3) if nparts > 1 else 0, "f32": int((nparts - 1) * npl * 12) if nparts > 1 else 0}
        del rgb, face
    return {"ms_per_frame_part0_of_N": {str(n): ms_med[n] for n in ms}, "ms_per_frame_part0_of_N_min": {str(n): ms[n] for n in ms},
            "render_speedup_bound": {str(n): ms_med[1] / ms_med[n] for n in ms}, "render_speedup_bound_of_min": {str(n): ms[1] / ms[n] for n in ms},
            "frames_per_call": {str(n): batch_of[n] for n in ms}, "calls_in_flight": inflight,
            "kernel_span_ms_per_frame_part0_of_N": {str(n): spans[n] for n in ms},
            "gather_bytes_into_rank0_per_frame": {str(n): gather[n] for n in ms},
            "note": "one GPU renders rank 0's tiles of an N-rank run (same tile size, same round-robin, min(64, %d N) frames per call, camera path); " % per_rank +
                    "median (and min) of three timings of >= 128 frames; ms[1] / ms[N] bounds the N-GPU speed-up from the render side"}


def oracle_scene(wl):
    from oracle import oracle_api as oa
    oa.build()
    s = oa.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [oa.make_mtl(**m) for m in wl["mats"]])
    s.set_lights(wl["lights"])
    return oa, s


PMC_FILES = [os.path.join("profiles", "r03_pmc.json"), os.path.join("profiles", "r02_pmc.json")]   # scripts/prof_derive.py --pmc-json over a round's rocprofv3 passes (newest first)


def any_blocker_shadows(wl):
    """librtx stops a shadow ray at any blocker when every material shades a blocked light alike (rtx.hip, commit)."""
    m = wl["mats"]
    if not m:
        return False
    key = lambda d: tuple((k, tuple(np.ravel(d.get(k, dflt)).tolist())) for k, dflt in
                          (("ka", (.3, .3, .3)), ("kd", (.5, .5, .5)), ("ks", (1., 1., 1.)), ("ns", 10.0), ("d", 1.0)))
    tm = np.asarray(wl["tri_mat"])
    return all(key(x) == key(m[0]) for x in m) and bool((tm >= 0).all() and (tm < len(m)).all())


def shadow_rays_of(wl, o, d1, t, hit):
    """Hit points / shadow directions of the first light in numpy fp32 (ulp-level differences from the kernel's own values
    do not move the visit counts): (P, dir, tmax)."""
    dd = d1[hit]
    d2 = dd / np.sqrt((dd * dd).sum(1, dtype=np.float32))[:, None]
    P = (o[hit] + t[hit, None] * d2).astype(np.float32)
    L = np.float32(wl["lights"][0])
    sd = (L[None] - P).astype(np.float32)
    dist = np.sqrt((sd * sd).sum(1, dtype=np.float32)).astype(np.float32)
    return P, (sd / dist[:, None]).astype(np.float32), (dist * np.float32(1.001)).astype(np.float32)     # shadow_tmax(), rtx_kernels.h


def roofline(args, wl, scene, info, ms, rays, frames_per_launch=1, views=None):
    """The dominant traversal kernel against the rooflines that could bound it.

    ms: per-frame kernel durations measured one render call at a time (HIP events on the library's stream); a call's
    launches carry frames_per_launch frames (rtx_render_tiles_batch_dev, cameras `views`), and every per-launch figure
    below is for such a launch.
    * algorithmic_bytes_per_launch: SURVEY 8(d)'s figure, 32 B per node visited + 36 B per triangle tested + io,
      counted by the CPU mirror (oracle/rt_oracle_bvh.c) walking the same BVH in the same order on every k-th
      pixel's rays of those cameras (shadow rays bounded at the light and stopped at any blocker exactly where the
      kernel does).  It is LOAD DEMAND: nearly all of it is served by L1 / L2 / Infinity Cache, so demand / duration
      exceeds the HBM peak; it is printed as survey_8d_frac (the task's definition, non-physical above 1) and as
      `load_demand` against the L2, never as the HBM figure.
    * achieved / frac (bound "hbm"): the bytes that did cross the HBM-side interface per launch (`traffic`, PMC
      FETCH_SIZE x2 + WRITE_SIZE from the committed rocprofv3 passes of this command: traffic_source says which file;
      counters cannot be read inside this run) / the launch duration measured here, against 8 TB/s; without a PMC
      record, the compulsory bytes (scene + io touched once).
    * valu: the bound that binds -- VALU-busy share of the kernel's SIMD cycles x active lanes per VALU
      instruction (same PMC passes): the fraction of the vector ALU's lane-cycles doing useful work.
    """
    oa, osc = oracle_scene(wl)
    W, H = wl["W"], wl["H"]
    stride = max(1, args.roofline_stride)
    B = frames_per_launch
    views = list(views) if views else [wl["view"]]
    nodes, order = scene.bvh()
    tris = scene.tris()
    lo, hi = osc.root_box()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(16, cores)
    # the sample: every (stride * len(views))-th pixel of each camera of the launch, offsets staggered
    o_l, d_l = [], []
    for f, view in enumerate(views):
        ocam = oa.make_camera(view, W, H)
        o, d1 = oa.camera_rays(ocam, (0, 0, W, H))
        pick = np.arange((f * stride) % (stride * len(views)), W * H, stride * len(views))
        o_l.append(o[pick]); d_l.append(d1[pick])
    o_s, d_s = np.concatenate(o_l), np.concatenate(d_l)
    scale = float(W * H) / len(o_s)              # one frame's rays per sampled ray
    t, f, nv, tt, gs, tf = oa.bvh_trace(nodes, tris, order, lo, hi, o_s, d_s, threads=threads, nwild=info.nwild, layout_counts=True)
    prim_bytes = (32.0 * nv.sum() + 36.0 * tt.sum()) * scale + 8.0 * W * H
    # what the device layout really loads: 64-B packed groups, 16 B for a triangle's plane stage, 48 B more when it
    # goes on to the barycentrics, 8 B result
    prim_layout = (64.0 * gs.sum() + 16.0 * tt.sum() + 48.0 * tf.sum()) * scale + 8.0 * W * H
    entering = {"primary": int((gs > 0).sum() * scale), "shadow": 0}
    per_ray = {"primary_nodes": float(nv.mean()), "primary_tris": float(tt.mean()),
               "primary_nodes_of_rays_entering_bvh": float(nv[gs > 0].mean()) if (gs > 0).any() else 0.0}
    hit = f >= 0
    sh_bytes = sh_layout = 0.0
    anyhit = any_blocker_shadows(wl)
    if hit.any() and wl["opts"][4]:
        # first light only, scaled by the light count
        P, sd1, tmax = shadow_rays_of(wl, o_s, d_s, t, hit)
        _, _, nv2, tt2, gs2, tf2 = oa.bvh_trace_shadow(nodes, tris, order, lo, hi, P, sd1, tmax, anyhit, threads=threads, nwild=info.nwild)
        nl = len(wl["lights"])
        sh_bytes = ((32.0 * nv2.sum() + 36.0 * tt2.sum()) + 72.0 * len(P)) * scale * nl
        sh_layout = ((64.0 * gs2.sum() + 16.0 * tt2.sum() + 48.0 * tf2.sum()) + 72.0 * len(P)) * scale * nl
        entering["shadow"] = int((gs2 > 0).sum() * scale) * nl
        per_ray.update({"shadow_nodes": float(nv2.mean()), "shadow_tris": float(tt2.mean()),
                        "shadow_rays_stop_at_any_blocker": bool(anyhit)})
    spp = wl["spp"]
    # (a frame without child rays takes the library's direct mode: its shadow launch is k_trace_shadow_direct)
    o_ = wl["opts"]
    direct = (o_[0] == 0 or (o_[1] == 0 and o_[2] == 0)) and o_[3] == 0 and os.environ.get("RTX_DIRECT", "1") != "0"
    cand = [("k_trace_camera", prim_bytes * B, prim_layout * B, ms[0] / spp * B),
            ("k_trace_shadow_direct" if direct else "k_trace_queue", sh_bytes * B, sh_layout * B, ms[1] / spp * B)]
    if wl["opts"][0] > 0:     # reflection launches share k_trace_queue's time; only the camera kernel is attributable
        cand = cand[:1]
    name, nbytes, lbytes, kms = max(cand, key=lambda c: c[3])
    scene_bytes = 64.0 * (info.nnodes / 4.0) + 64.0 * info.ntris + 4.0 * info.ntris
    io_bytes = (8.0 * W * H if name == "k_trace_camera" else 40.0 * rays[1] / spp) * B
    compulsory = scene_bytes + io_bytes
    pmc, pmc_file = {}, None
    for cand_file in PMC_FILES:
        try:
            with open(os.path.join(ROOT, cand_file)) as fh:
                recs = json.load(fh).get(args.workload, {})
                rec = recs.get(name) or (recs.get("k_trace_queue", {}) if name == "k_trace_shadow_direct" else {})     # (records older than the direct mode)
        except (OSError, ValueError):
            continue
        if rec:
            pmc, pmc_file = rec, cand_file
            break
    traffic = pmc.get("hbm_bytes_per_launch")
    pmc_frames = pmc.get("frames_per_launch", 8)          # (records without the key: rounds 2-3, calls of 8 frames)
    if traffic is not None and pmc_frames != B:             # the record is for launches of another size
        traffic = traffic * B / pmc_frames
    sec = kms * 1e-3
    hbm_bytes = traffic if traffic is not None else compulsory
    res = {"bound": "hbm", "kernel": name, "unit": "GB/s", "peak": HBM_PEAK_GBS,
           "achieved": hbm_bytes / sec / 1e9 if sec > 0 else 0.0,
           "traffic": traffic, "avg_launch_ms": kms, "frames_per_launch": B,
           "traffic_source": ("committed: %s (rocprofv3 --pmc passes of this command in an earlier run; a PMC counter cannot be read inside the timed run)" % pmc_file)
                             if traffic is not None else None,
           "achieved_is": ("HBM-side bytes per launch (traffic: PMC FETCH_SIZE x2 + WRITE_SIZE, %s; FETCH_SIZE counts Infinity-Cache hits too, so this is L2-miss "
                           "traffic, an upper bound on HBM bytes) / the launch duration measured in this run" % pmc_file) if traffic is not None
                          else "compulsory bytes per launch (no PMC record for this workload) / measured launch duration",
           "compulsory_bytes_per_launch": compulsory,
           "algorithmic_bytes_per_launch": nbytes,
           "bytes_model": "SURVEY 8(d): 32 B/node visited + 36 B/triangle tested + io (8 B/primary, 72 B/queued ray); counts from the CPU "
                          "mirror traversal of the same BVH on every %d-th pixel's rays of the launch's %d cameras, scaled" % (stride, len(views)),
           "per_ray": per_ray}
    res["frac"] = res["achieved"] / HBM_PEAK_GBS
    # SURVEY 8(d)'s own figure: algorithmic bytes / launch duration / HBM peak.  Above 1 it says what it is: load demand that the
    # caches serve (the scene fits the Infinity Cache), not bytes that crossed the HBM interface.
    res["survey_8d_frac"] = nbytes / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else 0.0
    res["survey_8d_note"] = "algorithmic_bytes_per_launch / avg_launch_ms / 8 TB/s, the task's definition: non-physical above 1 (cache-served load demand); frac is the measured HBM-side figure"
    # the 8(d) bytes are load demand served by the caches: set against the L2's aggregate bandwidth, never against HBM
    L2_PEAK_GBS = 34500.0           # MI355X_MICROARCH.md: 8 x 4 MiB L2, ~34.5 TB/s aggregate
    res["load_demand"] = {"GB/s": nbytes / sec / 1e9 if sec > 0 else 0.0, "device_layout_GB/s": lbytes / sec / 1e9 if sec > 0 else 0.0,
                          "device_layout_bytes_per_launch": lbytes,
                          "frac_of_l2_peak": (lbytes / sec / 1e9 / L2_PEAK_GBS) if sec > 0 else 0.0, "l2_peak_GB/s": L2_PEAK_GBS,
                          "note": "served by L1 / L2 / Infinity Cache (scene resident: see traffic); not an HBM figure"}
    if traffic is not None:
        res["traffic_over_compulsory"] = traffic / compulsory
    if "valu_busy" in pmc and "active_lanes" in pmc:
        res["valu"] = {"bound": "valu", "busy": min(1.0, pmc["valu_busy"]), "active_lanes": pmc["active_lanes"],
                       "frac": min(1.0, pmc["valu_busy"]) * pmc["active_lanes"],
                       "issue_lower_bound": pmc.get("valu_issue_min"),
                       "frac_lower_bound": (pmc["valu_issue_min"] * pmc["active_lanes"]) if pmc.get("valu_issue_min") else None,
                       "waves_per_simd": pmc.get("waves_per_simd"), "ta_busy": pmc.get("ta_busy"), "td_busy": pmc.get("td_busy"),
                       "note": "the bound that binds: share of the kernel's SIMD cycles issuing vector ALU instructions (at most 1) x active lanes "
                               "per instruction (%s, rocprofv3 --pmc SQ passes of this command with --inflight 1: one call at a time)" % pmc_file}
        res["binding_bound"] = "valu"
    res["rays_entering_bvh_per_frame"] = {k: v * spp for k, v in entering.items()}
    # The same traversal of the same BVH on the host's cores, a WHOLE frame with its shadow rays (the "same algorithm on
    # the CPU" figure of SURVEY 8(d)): every pixel's primary ray of the first camera, then the shadow ray of every hit.
    ocam = oa.make_camera(views[0], W, H)
    o, d1 = oa.camera_rays(ocam, (0, 0, W, H))
    t0 = time.perf_counter()
    t, f, _, _ = oa.bvh_trace(nodes, tris, order, lo, hi, o, d1, threads=threads, nwild=info.nwild)
    t_prim = time.perf_counter() - t0
    n_sh, t_sh = 0, 0.0
    hit = f >= 0
    if hit.any() and wl["opts"][4]:
        P, sd1, tmax = shadow_rays_of(wl, o, d1, t, hit)
        t0 = time.perf_counter()
        oa.bvh_trace_shadow(nodes, tris, order, lo, hi, P, sd1, tmax, anyhit, threads=threads, nwild=info.nwild)
        t_sh = (time.perf_counter() - t0) * len(wl["lights"])
        n_sh = len(P) * len(wl["lights"])
    res["cpu_bvh_full_frame"] = {"value": (W * H + n_sh) / (t_prim + t_sh) / 1e6, "unit": "Mrays/s (primary + shadow)", "cores": threads, "host_cores": cores,
                                 "seconds": t_prim + t_sh, "rays": {"primary": W * H, "shadow": n_sh},
                                 "sample": "the whole frame of the first camera of the launch: %d primary rays + %d shadow rays%s, oracle/rt_oracle_bvh.c walking the "
                                           "library's own BVH" % (W * H, n_sh, " (first light timed, scaled by the light count)" if len(wl["lights"]) > 1 else "")}
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

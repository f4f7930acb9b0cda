"""Wave-scheduling study on the CPU (no GPU needed): logs the traversal steps of sampled 64-ray
groups of a bench workload with the CPU mirror (oracle/rt_oracle_bvh.c) and replays them under
different SIMD scheduling policies to estimate lane utilisation.

    python3 scripts/wave_sim.py [workload] [ngroups]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import bench  # noqa: E402
from oracle import oracle_api as oa  # noqa: E402
from raytracing_project_amd import api, scenes  # noqa: E402

CN, CL, CT = 110.0, 12.0, 62.0   # VALU instructions: ~110 per node-group step, ~62 per triangle test (ISA counts; the kernels are VALU-bound)


def events(nodes, tris, order, lo, hi, o, d1, nwild, stride=1024):
    n = len(o)
    t = np.zeros(n, np.float32); f = np.zeros(n, np.int32)
    ev = np.zeros((n, stride), np.uint8); nev = np.zeros(n, np.uint32)
    nodes = np.ascontiguousarray(nodes, np.uint32); tris = np.ascontiguousarray(tris, np.float32)
    order = np.ascontiguousarray(order, np.int32)
    o = np.ascontiguousarray(o, np.float32); d1 = np.ascontiguousarray(d1, np.float32)
    lo = np.ascontiguousarray(lo, np.float32); hi = np.ascontiguousarray(hi, np.float32)
    u32 = C.POINTER(C.c_uint32)
    oa.lib().ora_bvh_trace_events(nodes.ctypes.data_as(u32), oa.fp(tris), order.ctypes.data_as(C.POINTER(C.c_int32)),
                                  len(order), nwild, oa.fp(lo), oa.fp(hi), oa.fp(o), oa.fp(d1), n, oa.fp(t),
                                  f.ctypes.data_as(C.POINTER(C.c_int32)), ev.ctypes.data_as(C.POINTER(C.c_uint8)), stride,
                                  nev.ctypes.data_as(u32))
    assert nev.max() <= stride
    return ev & 0x7f, nev, t, f


STAT = {}


def sim(groups, refill_min, defer_leaves=0, split_tris=False, tag=None):
    """groups: list of consecutive 64-ray groups a wave processes (each = list of 64 event arrays).
    refill_min: idle lanes needed before new rays are taken (65 = never, i.e. group by group).
    defer_leaves: lanes at a leaf wait until that many lanes are at leaves (or nothing else to do).
    Returns (useful lane-instructions, issued wave-instructions * 64)."""
    pending = [r for g in groups for r in g]
    if split_tris:
        pending = [np.concatenate([[0] if e == 0 else [1] * int(e) for e in r]).astype(np.uint8) if len(r) else r for r in pending]
    st = STAT.setdefault(tag, [0.0, 0.0, 0.0, 0.0]) if tag else [0.0] * 4
    pending.reverse()
    lanes = [None] * 64   # [events, pos]
    useful = 0.0
    issued = 0.0
    while True:
        idle = [i for i in range(64) if lanes[i] is None]
        nact = 64 - len(idle)
        if pending and (len(idle) >= refill_min or nact == 0):
            if refill_min > 64:
                # plain group mode: take a whole group
                for i in range(64):
                    if pending:
                        e = pending.pop()
                        lanes[i] = [e, 0] if len(e) else None
            else:
                for i in idle:
                    while pending:
                        e = pending.pop()
                        if len(e):
                            lanes[i] = [e, 0]
                            break
            issued += 20.0   # refill bookkeeping
        act = [l for l in lanes if l is not None]
        if not act:
            if not pending:
                break
            continue
        at_node = [l for l in act if l[0][l[1]] == 0]
        at_leaf = [l for l in act if l[0][l[1]] != 0]
        run_leaf = at_leaf and (len(at_leaf) >= defer_leaves or not at_node)
        if at_node:
            issued += CN
            useful += CN * len(at_node) / 64.0
            st[0] += CN; st[1] += CN * len(at_node) / 64.0
            for l in at_node:
                l[1] += 1
        if run_leaf:
            mx = max(int(l[0][l[1]]) for l in at_leaf)
            issued += CL + CT * mx
            st[2] += CL + CT * mx
            for l in at_leaf:
                useful += (CL + CT * int(l[0][l[1]])) / 64.0
                st[3] += (CL + CT * int(l[0][l[1]])) / 64.0
                l[1] += 1
        for i in range(64):
            l = lanes[i]
            if l is not None and l[1] >= len(l[0]):
                lanes[i] = None
    return useful, issued


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "soup1M_4k_primary_shadow"
    nwaves = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    per_wave = 8    # consecutive groups one simulated wave processes
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
    rng = np.random.RandomState(1)
    res = {}
    tot_ev = 0
    for w in range(nwaves):
        # a wave's groups: 8x8 pixel tiles; consecutive groups = the 4 sub-tiles of 16x16 blocks
        bx = rng.randint(0, W // 16 - 1); by = rng.randint(0, H // 16)
        groups = []
        for g in range(per_wave):
            blk = g // 4; sub = g % 4
            x0 = (bx + blk) * 16 + (sub & 1) * 8; y0 = by * 16 + (sub >> 1) * 8
            if x0 + 8 > W:
                continue
            o, d1 = oa.camera_rays(ocam, (x0, y0, x0 + 8, y0 + 8))
            ev, nev, t, f = events(nodes, tris, order, lo, hi, o, d1, info.nwild)
            groups.append([ev[i, :nev[i]].copy() for i in range(64)])
            tot_ev += int(nev.sum())
        for key, (rm, dl, *rest) in {"group": (65, 0), "refill32": (32, 0), "refill16": (16, 0), "refill8": (8, 0), "refill1": (1, 0),
                              "refill16+defer16": (16, 16), "refill16+defer32": (16, 32), "group+defer16": (65, 16),
                              "group+split": (65, 0, True), "refill16+split": (16, 0, True), "refill16+defer16+split": (16, 16, True),
                              "refill8+defer24+split": (8, 24, True)}.items():
            u, i = sim(groups, rm, dl, *rest, tag=key)
            a = res.setdefault(key, [0.0, 0.0])
            a[0] += u; a[1] += i
    print("workload %s: %d waves x %d groups, %.1f steps/ray" % (name, nwaves, per_wave, tot_ev / (nwaves * per_wave * 64.0)))
    base = res["group"][1]
    for k, (u, i) in res.items():
        st = STAT[k]
        print("  %-24s utilisation %.3f   issued %.3f of group mode   node: %.2f of issued at util %.2f, leaf: %.2f at util %.2f" % (
            k, u / i, i / base, st[0] / i, st[1] / max(st[0], 1), st[2] / i, st[3] / max(st[2], 1)))


if __name__ == "__main__":
    main()

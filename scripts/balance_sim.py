"""Work-distribution study on the CPU (no GPU needed): per-unit costs of a bench workload's camera
rays from the CPU mirror (oracle/rt_oracle_bvh.c: group steps and triangle tests of 6 sampled rays
per 8x8 tile), replayed under the unit-to-wave assignments the kernels could use.

    python3 scripts/balance_sim.py [workload] [--variants]

Prints max/mean of the per-wave (or per-block) finish times.  DESIGN.md section 5 quotes it.
"""
import heapq
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle_api as oa  # noqa: E402
from raytracing_project_amd import api, scenes  # noqa: E402

NCU = 256


def unit_costs(name, samples=6):
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
    bx = (W + 15) // 16
    by = (H + 15) // 16
    units = bx * by * 4                      # unit u = 4 * (16x16 block) + 8x8 sub-tile, as k_trace_camera numbers them
    ub, us = np.arange(units) // 4, np.arange(units) % 4
    x0 = (ub % bx) * 16 + (us & 1) * 8
    y0 = (ub // bx) * 16 + (us >> 1) * 8
    rng = np.random.default_rng(0)
    cost = np.zeros(units)
    for _ in range(samples):
        px = np.minimum(x0 + rng.integers(0, 8, units), W - 1)
        py = np.minimum(y0 + rng.integers(0, 8, units), H - 1)
        idx = py * W + px
        _, _, _, tt, gs, _ = oa.bvh_trace(nodes, tris, order, lo, hi, o[idx], d1[idx], threads=8, nwild=info.nwild,
                                          layout_counts=True)
        cost += gs * 1.0 + tt * 0.6 + 0.3    # group steps + triangle tests + setup, in units of one group step
    return cost / samples


def static_lists(cost, wpb, bpc):
    """wave w takes units w, w + W, ..."""
    nw = NCU * bpc * wpb
    tot = np.zeros(nw)
    np.add.at(tot, np.arange(len(cost)) % nw, cost)
    return tot.max() / tot.mean()


def block_pool(cost, wpb, bpc):
    """block b owns the unit quads b, b + B, ...; its waves take the next unit when free."""
    nb = NCU * bpc
    quad = np.arange(len(cost)) // 4
    ends = []
    for b in range(nb):
        h = [0.0] * wpb
        for c in cost[(quad % nb) == b]:
            heapq.heappush(h, heapq.heappop(h) + c)
        ends.append(max(h))
    return max(ends) / (cost.sum() / (nb * wpb))


def variants(cost, bx=240):
    """Other static dealings of the unit quads (`--variants`): block-pool finish time / mean wave time, block sums max/mean,
    sums per CU max/mean (block b runs on XCD b mod 8, CU (b / 8) mod 32)."""
    import math
    nq = len(cost) // 4
    qcost = cost.reshape(nq, 4)
    q = np.arange(nq)

    def evaluate(label, block_of_quad, wpb=4, bpc=7):
        nb = NCU * bpc
        order = np.argsort(block_of_quad, kind="stable")
        starts = np.searchsorted(block_of_quad[order], np.arange(nb + 1))
        ends = np.zeros(nb); tot = np.zeros(nb)
        for b in range(nb):
            h = [0.0] * wpb
            mine = order[starts[b]:starts[b + 1]]
            for qq in mine:
                for c in qcost[qq]:
                    heapq.heappush(h, heapq.heappop(h) + c)
            ends[b] = max(h); tot[b] = qcost[mine].sum()
        cu = (np.arange(nb) % 8) * 32 + (np.arange(nb) // 8) % 32
        cusum = np.zeros(NCU); np.add.at(cusum, cu % NCU, tot)
        print("  %-52s pool %.3f  block sums %.3f  per CU %.3f" % (label, ends.max() / (cost.sum() / (nb * wpb)), tot.max() / tot.mean(),
                                                                  cusum.max() / cusum.mean()))
    nb = NCU * 7
    evaluate("now: quad q -> block q mod B, 256 threads x 7", q % nb)
    for m in (263, 2437):
        if math.gcd(m, nq) == 1:
            evaluate("quad index times %d mod the number of quads" % m, ((q * pow(m, -1, nq)) % nq) % nb)
    x, y = q % bx, q // bx
    for a, c in ((1, 463), (13, 463)):
        evaluate("lattice b = (%d x + %d y) mod B" % (a, c), (x * a + y * c) % nb)
    evaluate("random dealing", np.random.default_rng(1).permutation(nq) % nb)
    o = np.argsort(-qcost.sum(1)); r = np.arange(nq)
    bo = np.empty(nq, int); bo[o] = np.where((r // nb) % 2 == 0, r % nb, nb - 1 - r % nb)
    evaluate("dealt by known cost (descending, snake)", bo)
    evaluate("448 threads x 4 per CU", q % (NCU * 4), 7, 4)
    evaluate("896 threads x 2 per CU", q % (NCU * 2), 14, 2)
    evaluate("one pool per CU (28 waves)", q % NCU, 28, 1)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = args[0] if args else "soup1M_4k_primary_shadow"
    cost = unit_costs(name)
    if "--variants" in sys.argv:
        print("workload %s: other static dealings of the unit quads" % name)
        variants(cost)
        return
    print("workload %s: %d units, mean cost %.1f group steps per ray" % (name, len(cost), cost.mean()))
    for bpc in (7, 8):
        print("  static per-wave lists, %d blocks/CU:  max/mean %.3f" % (bpc, static_lists(cost, 4, bpc)))
        print("  per-block pool (LDS counter), %d blocks/CU: max/mean %.3f" % (bpc, block_pool(cost, 4, bpc)))
    nw = NCU * 7 * 4
    tot = np.zeros(nw)
    np.add.at(tot, np.arange(len(cost)) % nw, cost)
    tw = tot.reshape(7, NCU, 4)
    print("  sums per SIMD max/mean %.3f, per CU %.3f (what a pool per CU / no tail at all would give)" % (
        tw.sum(0).max() / tw.sum(0).mean(), tw.sum((0, 2)).max() / tw.sum((0, 2)).mean()))


if __name__ == "__main__":
    main()

"""The C++ drop-in facade (raytracing_project_amd/host/flyscene.hpp) driven like the reference's
own callers (tests/cpp/facade_demo.cpp), checked against the oracle.  Exercises, through the
reference's method names: initialize (OBJ/MTL import), getIntersections, intersectTriangle,
intersectBox, canSeeLight, traceRay, ReflectDebugRay, raytraceScene (13 spheres, result.ppm)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_scene_fixture, oracle_materials

pytestmark = pytest.mark.gpu
DEMO = os.path.join(ROOT, "tests", "cpp", "facade_demo")


def write_obj(path, fx, names):
    with open(path + ".mtl", "w") as f:
        for n, m in zip(names, fx["mats"]):
            f.write("newmtl %s\nNs %.9g\nKa %.9g %.9g %.9g\nKd %.9g %.9g %.9g\nKs %.9g %.9g %.9g\nNi %.9g\nd %.9g\nillum %d\n\n" % (
                n, m[9], m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[10], m[11], int(m[12])))
    with open(path + ".obj", "w") as f:
        f.write("# written from tests/golden by test_gpu_facade.py\nmtllib %s.mtl\n" % os.path.basename(path))
        for v in fx["verts"]:
            f.write("v %.9g %.9g %.9g\n" % tuple(v))
        cur = None
        for tri, m in zip(fx["idx"], fx["tri_mat"]):
            if m != cur:
                f.write("usemtl %s\n" % names[m])
                cur = m
            f.write("f %d %d %d\n" % tuple(int(i) + 1 for i in tri))


def test_facade_matches_oracle(tmp_path, oa):
    from raytracing_project_amd import scenes
    assert os.path.exists(DEMO), "tests/cpp/facade_demo is built by __graft_entry__.build()"
    fx = load_scene_fixture("planeobj")
    names = ["mat%d" % i for i in range(len(fx["mats"]))]
    base = str(tmp_path / "scene")
    write_obj(base, fx, names)
    rng = np.random.default_rng(4)
    n = 40
    org = np.c_[rng.uniform(-1.5, 1.5, n), rng.uniform(0.4, 2.0, n), rng.uniform(-1.5, 2.5, n)].astype(np.float32)
    tgt = np.c_[rng.uniform(-1.2, 1.2, n), rng.uniform(0.0, 0.3, n), rng.uniform(-1.2, 1.2, n)].astype(np.float32)
    d = (tgt - org).astype(np.float32)
    d[::8] = np.float32([0, 1, 0])          # some rays into the sky
    with open(base + ".rays", "w") as f:
        for o_, d_ in zip(org, d):
            f.write("%.9g %.9g %.9g %.9g %.9g %.9g\n" % (*o_, *d_))
    W = 48
    # the mesh of the last frame: every vertex moved a little (Flyscene::updateVertices -> rtx_scene_update_vertices, a refit)
    moved = fx["verts"].astype(np.float32).copy()
    moved[:, 1] += (np.float32(0.08) * np.sin(np.float32(3.0) * moved[:, 0])).astype(np.float32)
    moved[:, 0] += np.float32(0.05)
    with open(base + ".moved", "w") as f:
        for v_ in moved:
            f.write("%.9g %.9g %.9g\n" % tuple(v_))
    out = subprocess.run([DEMO, base + ".obj", str(W), base + ".rays", base + ".moved"], cwd=str(tmp_path), capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l.split() for l in out.stdout.splitlines() if l.strip()]
    assert ["faces", "14", "materials", "2"] in lines
    assert [l for l in lines if l[0] == "center"][0][1:] == ["0", "0", "2"]

    o = oa.Scene(fx["verts"], fx["idx"], fx["tri_mat"], oracle_materials(fx["mats"]))
    o.set_lights([[-1.0, 1.0, 1.0]])
    assert o.build_leafboxes() == 4
    o.set_options()                          # the reference itself
    # the Ray the demo builds holds normalize(d); the facade hands that to the library, whose
    # Ray ctor normalises once more -- exactly what happens to a by-value Ray in the reference? No:
    # the reference passes the Ray object itself.  Feed the oracle the same double construction.
    d1 = np.zeros_like(d)
    tmp = np.zeros(3, np.float32)
    for i in range(n):
        oa.lib().ora_ray(oa.fp(org[i]), oa.fp(d[i]), oa.fp(tmp), oa.fp(d1[i]))
    h = o.closest(org, d1)
    rgb = o.trace(org, d1)
    vis, bm, _, _ = o.visible(np.tile(np.float32([-1, 1, 1]), (n, 1)), h["p"])
    start = max(i for i, l in enumerate(lines) if l and l[0] == "center") + 1
    it = iter(lines[start:])
    f32 = lambda xs: np.array([float(x) for x in xs], np.float32)
    for i in range(n):
        l = next(it)
        assert l[0] == "hit"
        assert np.float32(float(l[1])) == h["t"][i] or (np.isinf(float(l[1])) and np.isinf(h["t"][i]))
        assert int(l[2]) == h["face"][i] and int(l[3]) == h["mat"][i]
        assert np.array_equal(f32(l[4:7]), h["n"][i]) and np.array_equal(f32(l[7:10]), h["p"][i])
        l = next(it)
        assert l[0] == "tri" and int(l[2]) == h["face"][i]        # no spheres yet: triangles only, same winner
        l = next(it)
        want_box = oa.lib().ora_box(oa.fp(org[i]), oa.fp(d1[i]), oa.fp(np.float32([-2, 0, -2])), oa.fp(np.float32([2, .3, 2])))
        assert l[0] == "box" and (np.float32(float(l[1])) == np.float32(want_box))
        if h["t"][i] != np.inf:
            l = next(it)
            assert l[0] == "vis" and int(l[1]) == vis[i]
        l = next(it)
        assert l[0] == "rgb"
        assert np.allclose(f32(l[1:4]), rgb[i], rtol=1e-4, atol=1e-6)
    # the batched overloads (one launch for all rays) gave the single-ray answers, element for element
    l = next(it)
    assert l == ["batched", str(n), "0"]
    # the leaf list as `boxes`, the primitive classes through their virtual intersects(Ray&), trace_global_illum
    l = next(it)
    assert l[0] == "boxes" and int(l[1]) == 4 and [int(x) for x in l[2:]] == [len(f) for _, _, f in o.leaves()]
    h7 = np.zeros(7, np.float32)
    for i in range(10):
        l = next(it)
        oa.lib().ora_sphere(oa.fp(org[i]), oa.fp(d[i]), oa.fp(np.float32([0.1, 0.3, -0.2])), C.c_float(0.6), oa.fp(h7))
        assert l[0] == "isph" and int(l[2]) == -1 and np.array_equal(f32(l[1:2] + l[3:9]).view(np.uint32), h7.view(np.uint32))
        l = next(it)
        tv = np.float32([[-2, 0, -2], [2, 0, 2], [2, 0, -2]])
        oa.lib().ora_triangle(oa.fp(org[i]), oa.fp(d[i]), oa.fp(tv[0]), oa.fp(tv[1]), oa.fp(tv[2]), oa.fp(np.float32([0, 1, 0])), oa.fp(h7))
        assert l[0] == "itri" and np.array_equal(f32([l[1]] + l[3:6]).view(np.uint32), h7[[0, 4, 5, 6]].view(np.uint32))
        assert int(l[2]) == (0 if np.isfinite(h7[0]) else -1)
        l = next(it)
        want_box = oa.lib().ora_box(oa.fp(org[i]), oa.fp(d[i]), oa.fp(np.float32([-1, 0, -1])), oa.fp(np.float32([1, .5, 1])))
        assert l[0] == "ibox" and np.float32(float(l[1])) == np.float32(want_box)
        l = next(it)
        assert l[0] == "gi" and np.array_equal(f32(l[1:5]).view(np.uint32), o.trace_gi(org[i], d1[i])[0].view(np.uint32))
    # debug rays: ReflectDebugRay through 3 reflections for the first 12 rays (raw direction, as the demo passes it)
    for i in range(12):
        l = next(it)
        assert l[0] == "debug"
        nseg, nbox, nls = int(l[1]), int(l[2]), int(l[3])
        segs, faces, flags, ls = o.debug_ray(org[i], d[i], 3)
        assert nseg == len(segs) and nls == len(ls)
        for k in range(nseg):
            l = next(it)
            assert l[0] == "seg" and np.array_equal(f32(l[1:8]), segs[k, :7]) and int(l[8]) == int(segs[k, 7]) and int(l[9]) == faces[k]
        boxes = set()
        for k in range(nbox):
            l = next(it)
            assert l[0] == "hitbox"
            boxes.add(int(l[1]))
        assert boxes == set(np.nonzero(flags)[0].tolist())
        for k in range(nls):
            l = next(it)
            assert l[0] == "lseg" and np.array_equal(f32(l[1:8]), ls[k])
    # raytraceScene: planeobj + 13 spheres, reference options, result.ppm
    cr, ka, kd = scenes.reference_spheres()
    ids = [o.add_material(oa.make_mtl(ka=tuple(ka[i]), kd=tuple(kd[i]), ks=(.5, .5, .5), ns=10, ni=1, d=0, illum=2))
           for i in range(len(cr))]
    o.set_spheres(cr, ids)
    want, _, _ = o.render(oa.make_camera(scenes.default_view(), W, W), (0, 0, W, W), threads=8, want_hits=False)
    assert oa.lib().ora_ppm_write(str(tmp_path / "want.ppm").encode(), oa.fp(want), W, W, 0) == 0
    got_txt = open(tmp_path / "result_square.ppm").read().split()
    want_txt = open(tmp_path / "want.ppm").read().split()
    assert got_txt[:4] == want_txt[:4] == ["P3", str(W), str(W), "255"]
    g = np.array(got_txt[4:], np.int64)
    w = np.array(want_txt[4:], np.int64)
    assert g.shape == w.shape and np.abs(g - w).max() <= 1      # 1e-4 relative on [0,1] floats can move a byte by one
    assert (g != w).mean() < 0.01
    # result.ppm against the reference's OWN writer (Tucano::ImageImporter::writePPMImage compiled in place, oracle/_ref,
    # shipped prebuilt) fed the float frame the facade rendered: the same file, byte for byte
    frame = np.fromfile(tmp_path / "frame_square.bin", np.float32).reshape(W, W, 3)
    if oa.ref() is not None and hasattr(oa.ref(), "ref_write_ppm"):
        oa.ref_write_ppm(tmp_path / "ref_writer.ppm", frame)
        assert open(tmp_path / "ref_writer.ppm", "rb").read() == open(tmp_path / "result_square.ppm", "rb").read()
    assert ["frames", "1"] in lines          # the second updating_pixels call found the frame cached
    # a frame that is not square (the reference writes out of bounds there; the facade's [x][y] buffer has the
    # frame's own shape), rendered once by one scene handle and once as three parts of the tile grid on three handles
    # (rtx_render_multi through Flyscene::raytraceScene): the same file both times, equal to the oracle's frame
    W2, H2 = W + W // 2, W - W // 4
    want2, _, _ = o.render(oa.make_camera(scenes.default_view(), W2, H2), (0, 0, W2, H2), threads=8, want_hits=False)
    assert oa.lib().ora_ppm_write(str(tmp_path / "want_wide.ppm").encode(), oa.fp(want2), W2, H2, 0) == 0
    wide = open(tmp_path / "result_wide.ppm").read()
    assert wide == open(tmp_path / "result_wide_3parts.ppm").read()
    got_txt, want_txt = wide.split(), open(tmp_path / "want_wide.ppm").read().split()
    assert got_txt[:4] == want_txt[:4] == ["P3", str(W2), str(H2), "255"]
    g = np.array(got_txt[4:], np.int64)
    w = np.array(want_txt[4:], np.int64)
    assert g.shape == w.shape and np.abs(g - w).max() <= 1 and (g != w).mean() < 0.01
    assert ["nonsquare", "refused"] in lines
    # the moved mesh: the refitted scene's frame equals the oracle's on the new vertices (reference options: GI probes over the
    # leaf list split again from the new vertices, spheres, refraction)
    o2 = oa.Scene(moved, fx["idx"], fx["tri_mat"], oracle_materials(fx["mats"]))
    o2.set_lights([[-1.0, 1.0, 1.0]])
    nl2 = o2.build_leafboxes()
    o2.set_options()
    ids2 = [o2.add_material(oa.make_mtl(ka=tuple(ka[i]), kd=tuple(kd[i]), ks=(.5, .5, .5), ns=10, ni=1, d=0, illum=2)) for i in range(len(cr))]
    o2.set_spheres(cr, ids2)
    assert ["moved", str(len(moved)), "boxes", str(nl2)] in lines
    want3, _, _ = o2.render(oa.make_camera(scenes.default_view(), W2, H2), (0, 0, W2, H2), threads=8, want_hits=False)
    assert oa.lib().ora_ppm_write(str(tmp_path / "want_moved.ppm").encode(), oa.fp(want3), W2, H2, 0) == 0
    got_txt, want_txt = open(tmp_path / "result_moved.ppm").read().split(), open(tmp_path / "want_moved.ppm").read().split()
    g = np.array(got_txt[4:], np.int64)
    w = np.array(want_txt[4:], np.int64)
    assert g.shape == w.shape and np.abs(g - w).max() <= 1 and (g != w).mean() < 0.01
    assert (w != np.array(open(tmp_path / "want_wide.ppm").read().split()[4:], np.int64)).mean() > 0.01     # (the mesh did move)

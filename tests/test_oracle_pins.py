"""The CPU oracle against what pins it (CPU only):
  * golden vectors produced by the reference's OWN code (oracle/_ref) -- tests/golden/primitives.npz
  * the live reference build, where oracle/_ref/libref.so exists
  * the reference outputs recorded in SURVEY.md 8(c) (full flyscene.cpp run)
  * which bundled scenes divideBox_KD terminates on, and with how many leaves
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCENES, CAMS, assert_bits_equal, load_scene_fixture, oracle_materials

fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
f32 = lambda a: np.ascontiguousarray(a, np.float32)


@pytest.fixture(scope="module")
def prim():
    z = np.load(os.path.join(GOLDEN, "primitives.npz"))
    return {k: z[k] for k in z.files}


def _run_primitives(L, prefix, g):
    n = len(g["o"])
    out = {k: [] for k in ("tri", "plane", "box", "sphere", "ray", "reflect", "refract", "resend", "nface", "nq")}
    h = np.zeros(7, np.float32)
    a = np.zeros(3, np.float32)
    b = np.zeros(3, np.float32)
    F = lambda name: getattr(L, prefix + name)
    F("box").restype = C.c_float
    for i in range(n):
        o, d, v, nq, p = f32(g["o"][i]), f32(g["d"][i]), f32(g["v"][i]), f32(g["nq"][i]), f32(g["p"][i])
        nf = np.zeros(3, np.float32)
        F("face_normal")(fp(v[0]), fp(v[1]), fp(v[2]), fp(nf))
        out["nface"].append(nf.copy())
        q = np.zeros(3, np.float32)
        F("renormal")(fp(nf), fp(q))
        out["nq"].append(q.copy())
        F("triangle")(fp(o), fp(d), fp(v[0]), fp(v[1]), fp(v[2]), fp(nq), fp(h))
        out["tri"].append(h.copy())
        F("plane")(fp(o), fp(d), fp(v[0]), fp(nq), fp(h))
        out["plane"].append(h.copy())
        out["box"].append(F("box")(fp(o), fp(d), fp(f32(g["bmin"][i])), fp(f32(g["bmax"][i]))))
        ds = f32(g["sphere"][i][:3])
        F("sphere")(fp(o), fp(ds), fp(f32(g["c"][i])), C.c_float(float(g["r"][i])), fp(h))
        out["sphere"].append(np.r_[ds, h])
        F("ray")(fp(o), fp(d), fp(a), fp(b))
        out["ray"].append(np.r_[a, b])
        F("reflect")(fp(o), fp(d), fp(nq), fp(p), fp(a), fp(b))
        out["reflect"].append(np.r_[a, b])
        F("refract")(fp(o), fp(d), fp(nq), fp(p), fp(f32(g["view"][i])), C.c_float(float(g["n12"][i][0])),
                     C.c_float(float(g["n12"][i][1])), fp(a), fp(b))
        out["refract"].append(np.r_[a, b])
        dd = np.zeros(24, np.float32)
        F("resend")(fp(o), fp(d), fp(nq), fp(p), 8, fp(dd))
        out["resend"].append(dd)
    return {k: np.asarray(v, np.float32) for k, v in out.items()}


def test_oracle_matches_reference_golden_primitives(oa, prim):
    got = _run_primitives(oa.lib(), "ora_", prim)
    for k in got:
        assert_bits_equal(got[k], prim[k], k)


def test_oracle_camera_matches_reference_golden(oa, prim):
    L = oa.lib()
    for i in range(len(prim["view34"])):
        W, H = prim["wh"][i]
        cam = oa.make_camera(prim["view34"][i], W, H)
        w = np.zeros(3, np.float32)
        c = np.zeros(3, np.float32)
        L.ora_camera_point(C.byref(cam), float(prim["ij"][i][0]), float(prim["ij"][i][1]), fp(w), fp(c))
        assert_bits_equal(w, prim["cam_world"][i], "world %d" % i)
        assert_bits_equal(c, prim["cam_center"][i], "center %d" % i)


def test_oracle_matches_live_reference_build(oa, prim):
    R = oa.ref()
    if R is None:
        pytest.skip("oracle/_ref/libref.so not built (no /root/reference on this machine)")
    # fresh seed, not the committed vectors
    rng = np.random.default_rng(7)
    g = {k: v[:400].copy() for k, v in prim.items()}
    g["o"] = f32(rng.uniform(-3, 3, g["o"].shape))
    g["d"] = f32(rng.uniform(-1, 1, g["d"].shape))
    ref = _run_primitives(R, "ref_", g)
    g["nq"] = ref["nq"]
    ref = _run_primitives(R, "ref_", g)
    got = _run_primitives(oa.lib(), "ora_", g)
    for k in got:
        assert_bits_equal(got[k], ref[k], k)


def _default_scene(oa):
    fx = load_scene_fixture("planeobj")
    s = oa.Scene(fx["verts"], fx["idx"], fx["tri_mat"], oracle_materials(fx["mats"]))
    s.set_lights([[-1.0, 1.0, 1.0]])
    return s, fx


def test_is_in_box_matches_live_reference_build(oa):
    """Box::isInBox (src/box.h:87-93; the gate of the debug ray's leaf scan and of divideBox_KD's face assignment):
    closed on every side, incl. points exactly on a face, an edge, a corner, and NaN coordinates."""
    R = oa.ref()
    if R is None:
        pytest.skip("oracle/_ref was not built (no /root/reference here)")
    L = oa.lib()
    L.ora_is_in_box.restype = int
    rng = np.random.default_rng(12)
    bmin, bmax = np.float32([-2, 0, -2]), np.float32([2, 0.3, 2])
    pts = rng.uniform(-2.5, 2.5, (4000, 3)).astype(np.float32)
    pts[:, 1] = rng.uniform(-0.2, 0.5, 4000).astype(np.float32)
    k = rng.integers(0, 3, 1500)
    pts[np.arange(1500), k] = np.where(rng.random(1500) < 0.5, bmin[k], bmax[k])          # exactly on a face
    pts[1500:1600] = np.where(rng.random((100, 3)) < 0.5, bmin, bmax)                         # corners
    pts[1600:1650, 0] = np.nan
    pts[1650:1700] = np.nextafter(bmax, np.float32(np.inf))                                    # one ulp outside
    inside = 0
    for p in pts:
        a = R.ref_is_in_box(oa.fp(bmin), oa.fp(bmax), oa.fp(p))
        b = L.ora_is_in_box(oa.fp(bmin), oa.fp(bmax), oa.fp(p))
        assert a == b, p
        inside += a
    assert 500 < inside < 3500


def test_survey_recorded_reference_outputs(oa):
    """the only outputs of the complete reference (flyscene.cpp) we hold."""
    from raytracing_project_amd import scenes
    obs = json.load(open(os.path.join(GOLDEN, "survey_observations.json")))
    s, fx = _default_scene(oa)
    assert len(fx["idx"]) == obs["faces"] and len(fx["verts"]) == obs["verts"] and len(fx["mats"]) == obs["mats"]
    lo, hi = s.root_box()
    assert np.allclose(lo, obs["root_min"]) and np.allclose(hi, obs["root_max"])
    assert s.build_leafboxes() == obs["boxes"]
    cam = oa.make_camera(scenes.default_view(), 64, 64)
    rgb, t, face = s.render(cam, (0, 0, 64, 64))
    assert int((face >= 0).sum()) == obs["primaries_hitting_a_triangle"]
    assert t[32, 32] == obs["pixel_32_32"]["t"] and face[32, 32] == obs["pixel_32_32"]["faceId"]
    assert fx["tri_mat"][face[32, 32]] == obs["pixel_32_32"]["material_id"]
    assert np.allclose(rgb[40, 32], obs["traceRay_32_40"], rtol=1e-7)
    cr, ka, kd = scenes.reference_spheres()
    ids = [s.add_material(oa.make_mtl(ka=tuple(ka[i]), kd=tuple(kd[i]), ks=(.5, .5, .5), ns=10, ni=1, d=0, illum=2))
           for i in range(len(cr))]
    s.set_spheres(cr, ids)
    for W in (64, 128):
        cam = oa.make_camera(scenes.default_view(), W, W)
        rgb, _, _ = s.render(cam, (W // 2, W // 2, W // 2 + 1, W // 2 + 1))
        assert np.allclose(rgb[0, 0], obs["with_spheres_center_pixel_64_and_128"], rtol=2e-7, atol=1e-9)


@pytest.mark.parametrize("name,leaves", [("planeobj", 4), ("cube", 2), ("twoObjects", 1), ("plane", 1), ("toy", 241),
                                         ("test", -1), ("spheres", -1), ("dodgeColorTest", -1)])
def test_leafbox_split_matches_survey_f13(oa, name, leaves):
    fx = load_scene_fixture(name)
    s = oa.Scene(fx["verts"], fx["idx"], fx["tri_mat"], oracle_materials(fx["mats"]))
    assert s.build_leafboxes() == leaves
    assert int(fx["leaves"]) == leaves


@pytest.mark.parametrize("name", ["planeobj", "cube", "twoObjects", "test", "plane", "spheres"])
def test_oracle_reproduces_scene_goldens(oa, name):
    """The committed dumps are what the oracle computes today (guards the fixtures)."""
    fx = load_scene_fixture(name)
    s = oa.Scene(fx["verts"], fx["idx"], fx["tri_mat"], oracle_materials(fx["mats"]))
    s.set_lights([[-1.0, 1.0, 1.0]])
    res = int(fx["res"])
    for cname in CAMS:
        cam = oa.make_camera(fx["view_" + cname], res, res)
        _, t, face = s.render(cam, (0, 0, res, res), threads=4, want_rgb=False)
        assert np.array_equal(face, fx["face_" + cname])
        assert_bits_equal(t, fx["t_" + cname], "t")
        if "rgb_ns_" + cname in fx:
            s.set_options(max_bounces=2, reflection=1, refraction=0, gi_rays=0, shadows=1, skip_zero_weight=0)
            rgb, _, _ = s.render(cam, (0, 0, res, res), threads=4, want_hits=False)
            assert_bits_equal(rgb, fx["rgb_ns_" + cname], "rgb")


def test_obj_loader_face_order_and_materials(oa, tmp_path):
    """Tucano face order: triangles grouped per usemtl block in file order; ids are 0-based;
    a face with 4 indices is flattened 3 at a time (objimporter.hpp:191-223, mesh.hpp:441-468)."""
    (tmp_path / "m.mtl").write_text("newmtl red\nKa 0.1 0.2 0.3\nKd 1 0 0\nKs 0.5 0.5 0.5\nNs 20\nNi 1.5\nd 0.25\nillum 2\n"
                                    "\nnewmtl blue\nKd 0 0 1\n")
    (tmp_path / "m.obj").write_text("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nv 1 1 1\nv 2 2 2\n"
                                    "usemtl blue\nf 1 2 3\nusemtl red\nf 2/1/1 3/2/2 4/3/3\nf 1//1 2//2 5//3 6//4\n"
                                    "usemtl blue\nf 4 5 6\nf 3 4 5\n")
    v, idx, tm, mats, names = oa.obj_load(str(tmp_path / "m.obj"))
    assert names == ["red", "blue"] and len(v) == 6
    # groups: blue{1 2 3} red{2 3 4, 1 2 5 6 -> flat} blue{4 5 6, 3 4 5}; red's 7 indices give 2 faces
    assert idx.tolist() == [[0, 1, 2], [1, 2, 3], [0, 1, 4], [3, 4, 5], [2, 3, 4]]
    assert tm.tolist() == [1, 0, 0, 1, 1]
    assert np.allclose(tuple(mats[0].ka), (0.1, 0.2, 0.3)) and mats[0].ns == 20 and mats[0].d == 0.25
    assert tuple(mats[1].ka) == pytest.approx((0.3, 0.3, 0.3))   # Mtl() defaults survive


def test_ppm_writer(oa, tmp_path):
    rgb = np.zeros((2, 2, 3), np.float32)
    rgb[0, 1] = (1.0, 0.5, 0.999)
    rgb[1, 0] = (2.0, 0.0, 0.25)
    p = str(tmp_path / "o.ppm")
    assert oa.lib().ora_ppm_write(p.encode(), fp(rgb), 2, 2, 0) == 0
    assert open(p).read() == "P3\n2 2\n255\n0 0 0 255 127 254 \n255 0 63 0 0 0 \n"


# ---------------------------------------------------------------- the data formats either side of the path
# tests/golden/io_ref.json holds what the reference's OWN loadMTL / Mtl() / writePPMImage gave (oracle/_ref, generated by
# tests/golden/make_golden.py io): the oracle's restatements (oracle/rt_oracle_io.c) must reproduce them bit for bit.
@pytest.fixture(scope="module")
def io_ref():
    return json.load(open(os.path.join(GOLDEN, "io_ref.json")))


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _oracle_mtl(oa, text, tmp_path):
    with open(tmp_path / "m.mtl", "w", newline="") as fh:
        fh.write(text)
    (tmp_path / "m.obj").write_text("mtllib m.mtl\n")
    _, _, _, mats, names = oa.obj_load(str(tmp_path / "m.obj"))
    vals = np.array([[*m.ka, *m.kd, *m.ks, m.ns, m.ni, m.d, float(m.illum)] for m in mats], np.float32).reshape(-1, 13)
    return names, vals


def test_mtl_loader_matches_the_reference_executed_fixture(oa, io_ref, tmp_path):
    assert len(io_ref["mtl"]) >= 14 and sum(k.startswith("bundled:") for k in io_ref["mtl"]) == 7
    for key, rec in io_ref["mtl"].items():
        names, vals = _oracle_mtl(oa, rec["text"], tmp_path)
        assert names == rec["names"], key
        assert np.array_equal(_bits(vals).reshape(-1, 13), np.array(rec["values_bits"], np.uint32).reshape(-1, 13)), key


def test_default_mtl_matches_the_reference_executed_fixture(oa, io_ref, tmp_path):
    want = np.array(io_ref["default_mtl_bits"], np.uint32)
    assert np.array_equal(want.view(np.float32), np.float32([.3, .3, .3, .5, .5, .5, 1, 1, 1, 10, 0, 1, 0]))    # mtl.hpp:21-39
    m = oa.make_mtl()                                   # the Python-side default the tests build materials from
    assert np.array_equal(_bits([*m.ka, *m.kd, *m.ks, m.ns, m.ni, m.d, float(m.illum)]), want)
    _, vals = _oracle_mtl(oa, "", tmp_path)             # mtlIO.hpp:114-119: no material in the file -> one Mtl()
    assert np.array_equal(_bits(vals[0]), want)


def test_ppm_writer_matches_the_reference_executed_fixture(oa, io_ref, tmp_path):
    assert [r["n"] for r in io_ref["ppm"]] == [1, 2, 5, 16, 33]
    for rec in io_ref["ppm"]:
        n = rec["n"]
        rgb = np.array(rec["rgb_bits"], np.uint32).view(np.float32).reshape(n, n, 3)
        p = str(tmp_path / "o.ppm")
        assert oa.lib().ora_ppm_write(p.encode(), fp(rgb), n, n, 0) == 0
        assert open(p, newline="").read() == rec["text"], n
    assert any("-" in r["text"] for r in io_ref["ppm"])         # negative channels print as negative numbers there


def test_io_restatements_match_the_live_reference_build(oa, tmp_path):
    if oa.ref() is None or not hasattr(oa.ref(), "ref_load_mtl"):
        pytest.skip("oracle/_ref/libref.so not built with the I/O doors (no /root/reference on this machine)")
    rng = np.random.default_rng(99)
    for n in (3, 24):
        rgb = (rng.random((n, n, 3)) * 3 - 1).astype(np.float32)
        oa.ref_write_ppm(tmp_path / "r.ppm", rgb)
        assert oa.lib().ora_ppm_write(str(tmp_path / "o.ppm").encode(), fp(rgb), n, n, 0) == 0
        assert open(tmp_path / "o.ppm", "rb").read() == open(tmp_path / "r.ppm", "rb").read()
    keys = ["Ka", "Kd", "Ks", "Ns", "Ni", "d", "illum", "Tf", "#"]
    for it in range(40):
        lines = []
        for _ in range(int(rng.integers(1, 12))):
            k = keys[int(rng.integers(0, len(keys)))]
            if rng.random() < 0.25:
                lines.append("newmtl m%d" % len(lines))
            nums = " ".join("%.6g" % x for x in rng.normal(0, 2, 3 if k in ("Ka", "Kd", "Ks", "Tf") else 1))
            lines.append(k + (" " if rng.random() < 0.9 else "  ") + nums + (" " if rng.random() < 0.2 else ""))
        text = "newmtl first\n" + "\n".join(lines) + "\n"
        with open(tmp_path / "t.mtl", "w", newline="") as fh:
            fh.write(text)
        rnames, rvals = oa.ref_load_mtl(tmp_path / "t.mtl")
        names, vals = _oracle_mtl(oa, text, tmp_path)
        assert names == rnames, text
        assert np.array_equal(_bits(vals), _bits(rvals)), text

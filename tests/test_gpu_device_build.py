"""The hierarchy built and refitted on the GPU (RTX_FLAG_DEVICE_BUILD, rtx_scene_update_vertices): same hits as the
host-built tree and as the oracle's brute force, bit for bit -- a conservative hierarchy cannot change a (t, face id)
minimum, whichever builder made it.  Replaces Box(mesh) + divideBox_KD (src/box.h:41-85,
src/flyscene.cpp:77-82,1110-1185)."""
import numpy as np
import pytest

from conftest import SCENES, CAMS, assert_bits_equal, load_scene_fixture, oracle_materials, rtx_materials

pytestmark = pytest.mark.gpu

RGB_RTOL, RGB_ATOL = 1e-4, 1e-6      # north_star: "shaded RGB within 1e-4 relative"


@pytest.fixture(scope="module")
def api():
    from raytracing_project_amd import api
    assert api.lib().rtx_device_count() >= 1, "no GPU visible: the HIP path cannot run (there is no fallback)"
    return api


def opts(api, bounces=0):
    return api.Options(bounces, 1, 0, 0, 1, 0)


def one_material(api):
    return [api.make_material(ka=(0, 0, 0), kd=(.7, .7, .7), ks=(.2, .2, .2), ns=10, ni=1, d=1, illum=2)]


def make(api, verts, idx, tri_mat, mats, device_build, lights=((-2.0, 2.0, 2.0),), flags=0):
    s = api.Scene(verts, idx, tri_mat, mats)
    s.set_flags(flags | (api.FLAG_DEVICE_BUILD if device_build else 0))
    s.set_lights(lights)
    s.commit(0)
    return s


def check_tree(s, ntris):
    """The exported fp32 groups: every regular triangle in exactly one leaf, children inside their parents' slots,
    the packed planes outside the fp32 boxes, depth within the traversal stack."""
    i = s.info()
    nodes, order = s.bvh()
    packed = s.packed_bvh()
    tris = s.tris()
    assert sorted(order.tolist()) == list(range(ntris))
    q = nodes.reshape(-1, 4, 8)
    lo = q[:, :, 0:3].view(np.float32); hi = q[:, :, 4:7].view(np.float32)
    ref = q[:, :, 3]; cnt = q[:, :, 7]
    used = ~((ref == 0xFFFFFFFF) & (cnt == 0))
    seen = np.zeros(ntris, np.int32)
    depth_of = {0: 1}
    stack = [0]
    maxd = 0
    # padded boxes are not exported; a leaf's box must at least hold its triangles' vertices
    v0 = tris[:, 0:3]; v1 = v0 + tris[:, 4:7]; v2 = v0 + tris[:, 8:11]
    tlo = np.minimum(np.minimum(v0, v1), v2); thi = np.maximum(np.maximum(v0, v1), v2)
    while stack:
        g = stack.pop()
        d = depth_of[g]; maxd = max(maxd, d)
        for k in range(4):
            if not used[g, k]:
                continue
            if cnt[g, k]:
                a, n = int(ref[g, k]), int(cnt[g, k])
                assert n <= 4 and a + n <= ntris - i.nwild
                seen[a:a + n] += 1
                assert (tlo[a:a + n] >= lo[g, k]).all() and (thi[a:a + n] <= hi[g, k]).all()
            else:
                c = int(ref[g, k])
                cu = used[c]
                assert cu.any()
                assert (lo[c][cu] >= lo[g, k]).all() and (hi[c][cu] <= hi[g, k]).all(), "child group outside its slot"
                depth_of[c] = d + 1
                stack.append(c)
    assert (seen[: ntris - i.nwild] == 1).all() and (seen[ntris - i.nwild:] == 0).all()
    assert maxd == i.max_depth and 3 * maxd <= 48
    # packed encoding: origin + byte * step, rounded outwards
    P = packed
    org = P[:, 0:3].view(np.float32)
    step = np.stack([P[:, 3], P[:, 14], P[:, 15]], 1).view(np.float32)
    for a in range(3):
        for k in range(4):
            bl = ((P[:, 8 + a] >> (8 * k)) & 0xFF).astype(np.float32)
            bh = ((P[:, 11 + a] >> (8 * k)) & 0xFF).astype(np.float32)
            u = used[:, k]
            assert ((org[:, a] + bl * step[:, a])[u] <= lo[:, k, a][u]).all()
            assert ((org[:, a] + bh * step[:, a])[u] >= hi[:, k, a][u]).all()
    return maxd


def same_frames(api, a, b, cams, res, bounces=0, what=""):
    for cam in cams:
        ra = a.render(cam, opts(api, bounces), (0, 0, res[0], res[1]))
        rb = b.render(cam, opts(api, bounces), (0, 0, res[0], res[1]))
        assert np.array_equal(ra[2], rb[2]), what + ": face ids"
        assert_bits_equal(ra[1], rb[1], what + ": t")
        assert_bits_equal(ra[0], rb[0], what + ": rgb")
    return ra


@pytest.mark.parametrize("name", SCENES)
def test_device_built_scene_matches_golden(api, name):
    fx = load_scene_fixture(name)
    s = make(api, fx["verts"], fx["idx"], fx["tri_mat"], rtx_materials(fx["mats"]), True, lights=((-1.0, 1.0, 1.0),))
    ntris = len(fx["idx"])
    i = s.info()
    if ntris - i.nwild >= 8:
        assert i.build_kind == 1 and i.device_build_ms > 0
        check_tree(s, ntris)
    else:
        assert i.build_kind == 0            # too small: the host's builder
    res = int(fx["res"])
    for cname in CAMS:
        cam = api.make_camera(fx["view_" + cname], res, res)
        rgb, t, face = s.render(cam, api.Options(2, 1, 0, 0, 1, 0), (0, 0, res, res))
        assert np.array_equal(face, fx["face_" + cname]), "%s/%s face ids" % (name, cname)
        assert_bits_equal(t, fx["t_" + cname], "%s/%s t" % (name, cname))
        if "rgb_ns_" + cname in fx and name in ("planeobj", "cube", "twoObjects", "test", "spheres", "dodgeColorTest"):
            want = fx["rgb_ns_" + cname]
            assert (np.abs(rgb - want) <= RGB_ATOL + RGB_RTOL * np.abs(want)).all()
    s.close()


def test_records_and_wild_list_equal_the_host_builders(api):
    """Per-triangle constants (reference arithmetic), the wild list and the root gate's box computed on the GPU are the
    host's, bit for bit."""
    from raytracing_project_amd import scenes
    fx = load_scene_fixture("dodgeColorTest")        # has wild triangles
    for verts, idx, tm, mats in [(fx["verts"], fx["idx"], fx["tri_mat"], rtx_materials(fx["mats"])),
                                 (*scenes.soup(30000, seed=77, h=0.05), None, one_material(api))]:
        h = make(api, verts, idx, tm, mats, False)
        d = make(api, verts, idx, tm, mats, True)
        ih, idv = h.info(), d.info()
        assert idv.build_kind == 1 and ih.build_kind == 0 and ih.nwild == idv.nwild
        th, td = h.tris(), d.tris()
        oh, od = h.bvh()[1], d.bvh()[1]
        assert np.array_equal(np.sort(oh[len(oh) - ih.nwild:]), np.sort(od[len(od) - idv.nwild:]))
        bh = np.empty_like(th); bd = np.empty_like(td)
        bh[oh] = th; bd[od] = td
        assert_bits_equal(bh, bd, "triangle records")
        h.close(); d.close()


def test_soup_device_tree_against_host_tree_and_oracle(api, oa):
    from raytracing_project_amd import scenes
    n = 200000
    verts, idx = scenes.soup(n, seed=0x5EED0001, h=0.03)
    mats = one_material(api)
    h = make(api, verts, idx, None, mats, False)
    d = make(api, verts, idx, None, mats, True)
    assert d.info().build_kind == 1
    check_tree(d, n)
    cams = [api.make_camera(scenes.look_at_view((0, 0, 3), (0, 0, 0)), 480, 270),
            api.make_camera(scenes.look_at_view((0.3, 0.2, 0.1), (1, 0.5, -1)), 480, 270)]     # the second from inside
    same_frames(api, h, d, cams, (480, 270), what="soup")
    # incoherent rays, against the oracle's brute force
    rng = np.random.default_rng(5)
    m = 3000
    org = rng.uniform(-1.2, 1.2, (m, 3)).astype(np.float32)
    dr = rng.normal(size=(m, 3)).astype(np.float32)
    o = oa.Scene(verts, idx, None, [oa.make_mtl(ka=(0, 0, 0), kd=(.7, .7, .7), ks=(.2, .2, .2), ns=10, ni=1, d=1, illum=2)])
    ref = o.closest(org, dr, threads=8)
    t, face = d.trace_closest(org, dr)
    assert (ref["face"] >= 0).sum() > 1000
    assert np.array_equal(face, ref["face"])
    assert_bits_equal(t, ref["t"], "t")
    # every scheduler walks the device-built tree to the same hits
    for fl in (api.FLAG_SCHED_GROUPS, api.FLAG_SCHED_REFILL, api.FLAG_SCHED_REFILL | api.FLAG_SCHED_SPLIT):
        d.set_flags(api.FLAG_DEVICE_BUILD | fl)
        t2, f2 = d.trace_closest(org, dr)
        assert np.array_equal(f2, face)
        assert_bits_equal(t2, t, "t, scheduler 0x%x" % fl)
    h.close(); d.close()


def nasty_mesh(seed=3, n=4000):
    """Soup with thin and zero-area triangles, duplicates, a few vertices that are not numbers or beyond 1e18."""
    from raytracing_project_amd import scenes
    verts, idx = scenes.soup(n, seed=seed, h=0.2)
    verts = verts.copy()
    rng = np.random.default_rng(seed)
    v = verts.reshape(n, 3, 3)
    thin = rng.choice(n, 300, replace=False)
    q = 10.0 ** rng.uniform(-8, -2, 300)
    v[thin, 2] = v[thin, 0] + (v[thin, 1] - v[thin, 0]) * np.float32(0.7) + (np.sqrt(q)[:, None] * rng.normal(size=(300, 3)) * 0.2).astype(np.float32)
    v[thin[:20], 2] = v[thin[:20], 1]                       # zero area
    v[rng.choice(n, 5), 1, 0] = np.nan
    v[rng.choice(n, 5), 2, 1] = np.float32(3e38)
    v[rng.choice(n, 5), 0, 2] = np.inf
    v[100:110] = v[90:100]                                   # duplicated faces: the lower id wins
    return verts, idx


def test_nasty_mesh_device_tree_vs_oracle(api, oa):
    verts, idx = nasty_mesh()
    d = make(api, verts, idx, None, one_material(api), True)
    i = d.info()
    assert i.build_kind == 1 and i.nwild > 20
    o = oa.Scene(verts, idx, None, [oa.make_mtl(ka=(0, 0, 0), kd=(.7, .7, .7), ks=(.2, .2, .2), ns=10, ni=1, d=1, illum=2)])
    rng = np.random.default_rng(9)
    m = 20000
    org = rng.uniform(-1.5, 1.5, (m, 3)).astype(np.float32)
    dr = rng.normal(size=(m, 3)).astype(np.float32)
    dr[::7, 0] = 0.0
    ref = o.closest(org, dr, threads=8)
    t, face = d.trace_closest(org, dr)
    assert np.array_equal(face, ref["face"])
    assert_bits_equal(t, ref["t"], "t")
    d.close()


def wobble(verts, k):
    """A smooth deformation of the mesh (frame k of an animation)."""
    v = np.asarray(verts, np.float64)
    out = v.copy()
    out[:, 0] += 0.05 * k * np.sin(3.0 * v[:, 1] + 0.7 * k)
    out[:, 1] += 0.04 * k * np.cos(2.0 * v[:, 2] - 0.3 * k)
    out[:, 2] += 0.03 * k * np.sin(4.0 * v[:, 0] + 0.2 * k)
    return out.astype(np.float32)


@pytest.mark.parametrize("device_build", [False, True])
def test_refit_equals_a_fresh_scene(api, oa, device_build):
    from raytracing_project_amd import scenes
    verts, idx = scenes.blob(60, 61)
    mats = one_material(api)
    s = make(api, verts, idx, None, mats, device_build)
    cams = [api.make_camera(scenes.look_at_view((0, 0.4, 2.2), (0, 0, 0)), 320, 200),
            api.make_camera(scenes.look_at_view((1.5, 1.0, -1.0), (0, 0, 0)), 320, 200)]
    for k in (1, 2, 3):
        nv = wobble(verts, k)
        s.update_vertices(nv, api.UPDATE_REFIT)
        i = s.info()
        assert i.refits == k and i.refit_ms > 0
        check_tree(s, len(idx))
        fresh = make(api, nv, idx, None, mats, False)
        rgb, t, face = same_frames(api, s, fresh, cams, (320, 200), what="refit %d" % k)
        # ... and the untouched queries see the new mesh too
        rng = np.random.default_rng(k)
        org = rng.uniform(-1.5, 1.5, (5000, 3)).astype(np.float32)
        dr = rng.normal(size=(5000, 3)).astype(np.float32)
        ta, fa = s.trace_closest(org, dr); tb, fb = fresh.trace_closest(org, dr)
        assert np.array_equal(fa, fb)
        assert_bits_equal(ta, tb, "t")
        fresh.close()
    # the last frame against the oracle on the new vertices
    o = oa.Scene(nv, idx, None, [oa.make_mtl(ka=(0, 0, 0), kd=(.7, .7, .7), ks=(.2, .2, .2), ns=10, ni=1, d=1, illum=2)])
    o.set_lights([[-2.0, 2.0, 2.0]])
    o.set_options(max_bounces=0, reflection=1, refraction=0, gi_rays=0, shadows=1, skip_zero_weight=0)
    cam_o = oa.make_camera(scenes.look_at_view((1.5, 1.0, -1.0), (0, 0, 0)), 320, 200)
    orgb, ot, oface = o.render(cam_o, (100, 60, 196, 124), threads=8)
    assert np.array_equal(face[60:124, 100:196], oface)
    assert_bits_equal(t[60:124, 100:196], ot, "t vs oracle")
    assert (np.abs(rgb[60:124, 100:196] - orgb) <= RGB_ATOL + RGB_RTOL * np.abs(orgb)).all()
    # a rebuild on request, and vertices handed over on the device
    import torch
    nv = wobble(verts, 4)
    s.update_vertices(torch.from_numpy(nv).cuda(), api.UPDATE_REBUILD)
    assert s.info().refits == 0 and s.info().build_kind == (1 if device_build else 0)
    fresh = make(api, nv, idx, None, mats, False)
    # (what the info says about the rest of the scene outlives a rebuild)
    ia, ib = s.info(), fresh.info()
    assert (ia.nlights, ia.nmaterials, ia.nspheres, ia.ntris, ia.nverts) == (ib.nlights, ib.nmaterials, ib.nspheres, ib.ntris, ib.nverts)
    assert ia.nlights == 1 and ia.nmaterials == 1 and ia.device_bytes > 0
    same_frames(api, s, fresh, cams, (320, 200), what="rebuild")
    nv = wobble(verts, 5)
    s.update_vertices(torch.from_numpy(nv).cuda(), api.UPDATE_REFIT)
    fresh.close()
    fresh = make(api, nv, idx, None, mats, False)
    same_frames(api, s, fresh, cams, (320, 200), what="refit from a device pointer")
    # the host copy follows (exports, clone)
    c = s.clone(); c.commit(0)
    same_frames(api, c, fresh, cams[:1], (320, 200), what="clone after refit")
    c.close(); fresh.close(); s.close()


@pytest.mark.parametrize("device_build", [False, True])
def test_refit_that_degenerates_a_triangle_rebuilds(api, device_build):
    """A triangle of the hierarchy squeezed into a sliver (q below the wild threshold) cannot stay in it: the refit
    notices and the hierarchy is built again; hits are those of a fresh scene."""
    from raytracing_project_amd import scenes
    verts, idx = scenes.soup(5000, seed=21, h=0.1)
    mats = one_material(api)
    s = make(api, verts, idx, None, mats, device_build)
    assert s.info().nwild == 0
    nv = verts.copy().reshape(-1, 3, 3)
    for f in (17, 900, 4100):
        nv[f, 2] = nv[f, 0] + (nv[f, 1] - nv[f, 0]) * np.float32(0.5) + np.float32(1e-7)
    nv = nv.reshape(-1, 3)
    s.update_vertices(nv, api.UPDATE_REFIT)
    i = s.info()
    assert i.nwild >= 3 and i.refits == 0
    fresh = make(api, nv, idx, None, mats, False)
    cams = [api.make_camera(scenes.look_at_view((0, 0, 3), (0, 0, 0)), 256, 256)]
    same_frames(api, s, fresh, cams, (256, 256), what="degenerated")
    s.close(); fresh.close()


def test_hall_with_reflections_on_a_device_built_tree(api):
    """BASELINE config 3's scene (two materials: the closest blocker is searched; depth-3 reflections) both ways."""
    from raytracing_project_amd import scenes
    verts, idx, tm = scenes.hall(columns=8, seg=32, rings=32)
    mats = [api.make_material(ka=(.1, .1, .1), kd=(.6, .6, .6), ks=(.3, .3, .3), ns=20, ni=1, d=0.5, illum=2),
            api.make_material(ka=(.1, .05, .05), kd=(.7, .4, .3), ks=(.5, .5, .5), ns=30, ni=1, d=0.5, illum=2)]
    h = make(api, verts, idx, tm, mats, False, lights=((0.0, 2.5, 0.0),))
    d = make(api, verts, idx, tm, mats, True, lights=((0.0, 2.5, 0.0),))
    assert d.info().build_kind == 1
    cams = [api.make_camera(scenes.look_at_view((0, 1.5, 7.0), (0.5, 1.2, 0)), 320, 180)]
    same_frames(api, h, d, cams, (320, 180), bounces=3, what="hall")
    h.close(); d.close()


@pytest.mark.parametrize("device_build", [False, True])
def test_update_vertices_on_tiny_and_empty_meshes(api, oa, device_build):
    """Meshes below the GPU builder's size (the host's builder takes them: a lone leaf group, or nothing) refit and rebuild as well."""
    rng = np.random.default_rng(31)
    for n in (0, 1, 4, 9):
        T = rng.uniform(-1, 1, (n, 3, 3)).astype(np.float32)
        verts = T.reshape(-1, 3) if n else np.zeros((3, 3), np.float32)
        idx = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
        s = make(api, verts, idx, None, one_material(api), device_build)
        assert s.info().build_kind == (1 if device_build and n >= 8 else 0)
        org = rng.uniform(-2, 2, (4000, 3)).astype(np.float32)
        d = rng.normal(size=(4000, 3)).astype(np.float32)
        for how in (api.UPDATE_REFIT, api.UPDATE_REBUILD, api.UPDATE_REFIT):
            verts = (verts + rng.uniform(-0.2, 0.2, verts.shape)).astype(np.float32)
            s.update_vertices(verts, how)
            t, face = s.trace_closest(org, d)
            ref = oa.Scene(verts, idx, None, [oa.make_mtl()]).closest(org, d, threads=4)
            assert np.array_equal(face, ref["face"])
            assert_bits_equal(t, ref["t"], "t, %d triangles" % n)
        with pytest.raises(api.RtxError):
            s.update_vertices(np.zeros((len(verts) + 1, 3), np.float32))       # another vertex count is refused
        t, face = s.trace_closest(org, d)                                         # ... and leaves the scene as it was
        assert np.array_equal(face, ref["face"])
        s.close()


def test_fallbacks_of_the_gpu_builder(api, oa):
    """A collapse plan deeper than the traversal stack allows falls back to two binary levels per group, and that to the host's
    builder (test hooks lower the allowed depth, so that a small mesh gets there); runs of identical triangles (equal sort
    keys: subtrees balanced by index) and triangles that share one centre build fine."""
    import os
    from raytracing_project_amd import scenes
    verts, idx = scenes.soup(20000, seed=5, h=0.05)
    rng = np.random.default_rng(2)
    org = rng.uniform(-1.2, 1.2, (6000, 3)).astype(np.float32)
    d = rng.normal(size=(6000, 3)).astype(np.float32)
    o = oa.Scene(verts, idx, None, [oa.make_mtl()])
    ref = o.closest(org, d, threads=8)

    def build(env):
        os.environ.update(env)
        try:
            s = make(api, verts, idx, None, one_material(api), True)
        finally:
            for k in env:
                os.environ.pop(k, None)
        t, face = s.trace_closest(org, d)
        assert np.array_equal(face, ref["face"])
        assert_bits_equal(t, ref["t"], "t " + repr(env))
        i = s.info()
        s.close()
        return i
    plain = build({})
    assert plain.build_kind == 1
    fixed = build({"RTX_TEST_DEVICE_COLLAPSE": "fixed"})
    assert fixed.build_kind == 1 and fixed.nnodes != plain.nnodes              # the second rule made this tree
    lvl = build({"RTX_TEST_DEVICE_LEVELS": str(fixed.max_depth)})               # the plan (if deeper) gives way to the second rule
    assert lvl.build_kind == 1 and lvl.max_depth <= fixed.max_depth
    host = build({"RTX_TEST_DEVICE_LEVELS": "3"})                               # neither fits: the host's builder
    assert host.build_kind == 0
    # equal keys, shared centres
    tri = np.float32([[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    same = np.tile(tri, (3000, 1, 1))
    sizes = (10.0 ** rng.uniform(-3, 0, 3000)).astype(np.float32)
    nested = (tri[None] - np.float32([1 / 3, 1 / 3, 0])) * sizes[:, None, None] + np.float32([2, 2, 2])
    T = np.concatenate([same, nested]).astype(np.float32)
    v2, i2 = T.reshape(-1, 3), np.arange(3 * len(T), dtype=np.uint32).reshape(-1, 3)
    s = make(api, v2, i2, None, one_material(api), True)
    assert s.info().build_kind == 1
    check_tree(s, len(T))
    org2 = rng.uniform(-1, 3, (5000, 3)).astype(np.float32); org2[:, 2] = rng.uniform(0.5, 2.0, 5000)
    uv = rng.dirichlet([1, 1, 1], 5000)[:, :2] * 1.1 - 0.03                      # in and a little around the unit triangle at z = 0
    near = np.c_[2 + rng.uniform(-0.35, 0.35, 5000), 2 + rng.uniform(-0.35, 0.35, 5000), np.full(5000, 2.0)]   # ... and the nested ones at z = 2
    pick = rng.random(5000) < 0.5
    tgt = np.where(pick[:, None], np.c_[uv, np.zeros(5000)], near).astype(np.float32)
    t, face = s.trace_closest(org2, tgt - org2)
    r2 = oa.Scene(v2, i2, None, [oa.make_mtl()]).closest(org2, tgt - org2, threads=8)
    assert (r2["face"] >= 0).sum() > 1000
    assert np.array_equal(face, r2["face"])
    assert_bits_equal(t, r2["t"], "t, equal keys")
    s.close()


def test_fast_triangles_on_a_device_built_tree_and_after_a_refit(api):
    """RTX_FLAG_FAST_TRIANGLES reads the 48-byte records the GPU builder writes beside the reference ones: the same frames as the
    host-built scene with that flag (the opt-in test is not the reference's arithmetic, but it is the same arithmetic on either tree)."""
    from raytracing_project_amd import scenes
    verts, idx = scenes.blob(50, 51)
    mats = one_material(api)
    cams = [api.make_camera(scenes.look_at_view((0, 0.4, 2.2), (0, 0, 0)), 256, 160)]
    h = make(api, verts, idx, None, mats, False, flags=api.FLAG_FAST_TRIANGLES)
    d = make(api, verts, idx, None, mats, True, flags=api.FLAG_FAST_TRIANGLES)
    assert d.info().build_kind == 1
    same_frames(api, h, d, cams, (256, 160), what="fast triangles")
    nv = wobble(verts, 2)
    d.update_vertices(nv, api.UPDATE_REFIT)
    h2 = make(api, nv, idx, None, mats, False, flags=api.FLAG_FAST_TRIANGLES)
    same_frames(api, h2, d, cams, (256, 160), what="fast triangles after a refit")
    # the flag taken away again: the geometry is uploaded anew from the host copies of what the GPU built, reference arithmetic
    d.set_flags(api.FLAG_DEVICE_BUILD)
    d.commit(0)
    h3 = make(api, nv, idx, None, mats, False)
    same_frames(api, h3, d, cams, (256, 160), what="flag cleared")
    h.close(); d.close(); h2.close(); h3.close()

"""The input-domain guard and the verification mode.

The culling boxes of the hierarchy are padded for the rounding of o + t d with origins up to ~30 x the mesh's largest
|coordinate| away.  Rays from beyond are answered by a brute-force scan of every triangle on the GPU --
what the reference does for every ray -- so parity holds by construction; RTX_FLAG_VERIFY re-checks a sample of every
closest-hit launch the same way and reports mismatches in rtx_stats."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api():
    from raytracing_project_amd import api
    assert api.lib().rtx_device_count() >= 1, "no GPU visible: the HIP path cannot run (there is no fallback)"
    return api


def test_rays_from_far_outside_the_domain_match_brute_force(api, oa):
    from raytracing_project_amd import scenes
    verts, idx = scenes.soup(20000, seed=0x5EED0051, h=0.08)
    s = api.Scene(verts, idx, None, [api.make_material()])
    s.set_lights([[0.0, 3.0, 0.0]])
    s.commit(0)
    o = oa.Scene(verts, idx, None, [oa.make_mtl()])
    rng = np.random.default_rng(51)
    n = 6000
    T = verts.reshape(-1, 3, 3)
    target = (T[rng.integers(0, len(T), n)] * rng.dirichlet([1, 1, 1], n).astype(np.float32)[:, :, None]).sum(1)
    dirs = rng.normal(size=(n, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dist = 10.0 ** rng.uniform(0.5, 6.0, n)                      # 3 ... 1e6 scene extents away
    org = (target + dirs * dist[:, None]).astype(np.float32)
    d = (target - org).astype(np.float32)
    far = np.abs(org).max(1) > 30.0 * np.abs(verts).max()
    assert 0.5 < far.mean() < 0.95
    t, face = s.trace_closest(org, d)
    st = s.stats()
    h = o.closest(org, d, threads=16)
    assert (h["face"] >= 0).sum() > n // 20
    assert np.array_equal(face, h["face"])
    assert_bits_equal(t, h["t"], "t of far rays")
    assert st.rays_out_of_domain == int(far.sum())
    # canSeeLight from far away, and traceRay on far rays
    light = np.tile(np.float32([0, 3, 0]), (400, 1))
    vis, bm = s.trace_visible(light, org[:400])
    ovis, obm, _, _ = o.visible(light, org[:400])
    assert np.array_equal(vis, ovis) and np.array_equal(bm, obm)
    o.set_lights([[0.0, 3.0, 0.0]])
    o.set_options(1, 1, 0, 0, 1, 0)
    got = s.trace_shade(api.Options(1, 1, 0, 0, 1, 0), org[:600], d[:600])
    assert s.stats().rays_out_of_domain == int(far[:600].sum())
    want = o.trace(org[:600], d[:600])
    assert np.all(np.abs(got - want) <= 1e-6 + 1e-4 * np.abs(want))
    s.close()


def test_camera_beyond_the_domain(api, oa):
    from raytracing_project_amd import scenes
    verts, idx = scenes.soup(3000, seed=0x5EED0052, h=0.2)
    s = api.Scene(verts, idx, None, [api.make_material()])
    s.set_lights([[0.0, 3.0, 0.0]])
    s.commit(0)
    o = oa.Scene(verts, idx, None, [oa.make_mtl()])
    o.set_lights([[0.0, 3.0, 0.0]])
    o.set_options(0, 1, 0, 0, 1, 0)
    W = H = 48
    view = scenes.look_at_view((0.0, 0.0, 5000.0), (0, 0, 0))
    cam = api.make_camera(view, W, H, fovy=0.03)                 # a long lens: the soup fills the frame from 5000 away
    ocam = oa.make_camera(view, W, H, fovy=0.03)
    rgb, t, face = s.render(cam, api.Options(0, 1, 0, 0, 1, 0), (0, 0, W, H))
    assert s.stats().rays_out_of_domain == W * H
    orgb, ot, of = o.render(ocam, (0, 0, W, H), threads=16)
    assert (of >= 0).sum() > 200
    assert np.array_equal(face, of)
    assert_bits_equal(t, ot, "t from a camera 5000 away")
    assert np.all(np.abs(rgb - orgb) <= 1e-6 + 1e-4 * np.abs(orgb))
    s.close()
    # ... and on a mesh too big to scan per pixel the call says so instead of answering from boxes it cannot trust
    verts, idx = scenes.soup(400000, seed=0x5EED0053, h=0.03)
    s = api.Scene(verts, idx, None, [api.make_material()])
    s.set_lights([[0.0, 3.0, 0.0]])
    s.commit(0)
    with pytest.raises(api.RtxError, match="beyond what this mesh's culling boxes are padded for"):
        s.render(api.make_camera(view, 1920, 1080, fovy=0.03), api.Options(0, 1, 0, 0, 1, 0), (0, 0, 1920, 1080))
    s.close()


def test_verification_mode_counts_and_finds_what_it_should(api):
    from raytracing_project_amd import scenes
    verts, idx = scenes.soup(60000, seed=0x5EED0054, h=0.05)
    W, H = 640, 360
    cam = api.make_camera(scenes.look_at_view((0.2, 0.1, 2.8), (0, 0, 0)), W, H)
    rng = np.random.default_rng(54)
    org = rng.uniform(-1.5, 1.5, (50000, 3)).astype(np.float32)
    d = rng.normal(size=(50000, 3)).astype(np.float32)

    def run(pad_scale):
        if pad_scale is None:
            os.environ.pop("RTX_TEST_PAD_SCALE", None)
        else:
            os.environ["RTX_TEST_PAD_SCALE"] = str(pad_scale)
        try:
            s = api.Scene(verts, idx, None, [api.make_material()])
            s.set_lights([[-2.0, 2.0, 2.0]])
            s.set_flags(api.FLAG_VERIFY)
            s.commit(0)
        finally:
            os.environ.pop("RTX_TEST_PAD_SCALE", None)
        s.render(cam, api.Options(1, 1, 0, 0, 1, 0), (0, 0, W, H))
        a = s.stats()
        s.trace_closest(org, d)
        b = s.stats()
        s.set_flags(0)
        s.trace_closest(org, d)
        c = s.stats()
        s.close()
        return a, b, c

    a, b, c = run(None)
    assert 1500 < a.verify_checked <= 2200 and a.verify_mismatches == 0
    assert 1500 < b.verify_checked <= 2200 and b.verify_mismatches == 0
    assert c.verify_checked == 0                                  # off again: nothing is re-checked
    # boxes shrunk on purpose (test hook): the hierarchy now loses hits, and the verification must see it
    a, b, c = run(-300.0)
    assert a.verify_mismatches > 0 and b.verify_mismatches > 0


def test_guard_and_verification_inside_a_batch_of_frames(api):
    """rtx_render_tiles_batch_dev with one camera of the batch beyond the domain: its frame is answered by brute force (and only
    its pixels are counted), the other frames come from the hierarchy, each equal to its single call; with RTX_FLAG_VERIFY the
    sample is drawn over all frames of the batch."""
    import torch
    from raytracing_project_amd import scenes
    verts, idx = scenes.soup(3000, seed=0x5EED0052, h=0.2)
    s = api.Scene(verts, idx, None, [api.make_material()])
    s.set_lights([[0.0, 3.0, 0.0]])
    s.commit(0)
    W = H = 48
    near = api.make_camera(scenes.look_at_view((0.2, 0.1, 2.8), (0, 0, 0)), W, H)
    far = api.make_camera(scenes.look_at_view((0.0, 0.0, 5000.0), (0, 0, 0)), W, H, fovy=0.03)
    opts = api.Options(0, 1, 0, 0, 1, 0)
    rect = (0, 0, W, H)
    dev = torch.device("cuda", 0)
    P = W * H
    singles = []
    for cam in (near, far):
        rgb, t, face = s.render(cam, opts, rect)
        singles.append((rgb.reshape(P, 3), t.reshape(P), face.reshape(P)))
    order = [near, far, near]
    rgb = torch.zeros((3 * P, 3), dtype=torch.float32, device=dev)
    t = torch.zeros((3 * P,), dtype=torch.float32, device=dev)
    face = torch.full((3 * P,), -9, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    s.render_tiles_batch_dev(order, opts, rect, 0, 0, 1, rgb.data_ptr(), t.data_ptr(), face.data_ptr())
    st = s.stats()
    assert st.rays_out_of_domain == P                     # the far frame's pixels, nothing else
    for f, cam in enumerate(order):
        want = singles[0] if cam is near else singles[1]
        assert np.array_equal(face.cpu().numpy()[f * P:(f + 1) * P], want[2]), f
        assert_bits_equal(t.cpu().numpy()[f * P:(f + 1) * P], want[1], "t of frame %d" % f)
        assert_bits_equal(rgb.cpu().numpy()[f * P:(f + 1) * P], want[0], "rgb of frame %d" % f)
    assert (singles[1][2] >= 0).sum() > 200
    s.set_flags(api.FLAG_VERIFY)
    s.commit(0)
    big = torch.zeros((4 * P, 3), dtype=torch.float32, device=dev)
    bigf = torch.full((4 * P,), -9, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    s.render_tiles_batch_dev([near] * 4, opts, rect, 0, 0, 1, big.data_ptr(), None, bigf.data_ptr())
    st = s.stats()
    assert 1000 < st.verify_checked <= 4 * P and st.verify_mismatches == 0
    for f in range(4):
        assert np.array_equal(bigf.cpu().numpy()[f * P:(f + 1) * P], singles[0][2])
    s.close()

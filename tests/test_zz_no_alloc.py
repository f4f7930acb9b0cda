"""rtx_trace_closest_noalloc (the entry point that allocates nothing) against rtx_trace_closest_dev -- and, on the
second (smaller) call, against the CPU oracle's brute force over every triangle (the reference's own algorithm).

Collected last on purpose (file name): the round's end runs pytest -x.  Two calls in a row, the second with a
smaller input, so that what the first call left in the library's fixed scratch array (ray records, hit
records, stack overflow columns) would show if it leaked into the second; both bit-exact against the existing
entry point (face ids and t: integer / bit work, the bar of the other parity tests).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_no_alloc_entry_point_matches_trace_closest_dev():
    import torch
    from raytracing_project_amd import api, scenes
    assert api.lib().rtx_device_count() >= 1, "no GPU visible: the HIP path cannot run (there is no fallback)"
    # big, thin-walled triangles: deep traversal stacks, so the overflow columns in the scratch array are used too
    verts, idx = scenes.soup(30000, seed=0x5EED0042, h=0.5)
    s = api.Scene(verts, idx, None, [api.make_material()])
    s.set_lights([[0.0, 3.0, 0.0]])
    s.commit(0)
    rng = np.random.default_rng(42)
    dev = torch.device("cuda", 0)

    def rays(n):
        o = rng.uniform(-1.5, 1.5, (n, 3)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        return torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)

    def both(o, d):
        n = o.shape[0]
        t_a = torch.full((n,), -1.0, dtype=torch.float32, device=dev); f_a = torch.full((n,), -7, dtype=torch.int32, device=dev)
        t_b = torch.full((n,), -1.0, dtype=torch.float32, device=dev); f_b = torch.full((n,), -7, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        s.trace_closest_dev(o.data_ptr(), d.data_ptr(), n, t_a.data_ptr(), f_a.data_ptr())
        free0 = torch.cuda.mem_get_info()[0]
        s.trace_closest_noalloc(o.data_ptr(), d.data_ptr(), n, t_b.data_ptr(), f_b.data_ptr())
        assert torch.cuda.mem_get_info()[0] == free0, "device memory in use changed across the call"
        return t_a.cpu().numpy(), f_a.cpu().numpy(), t_b.cpu().numpy(), f_b.cpu().numpy()

    o1, d1 = rays(api.NOALLOC_MAX_RAYS)            # the maximum size
    o2, d2 = rays(4097)                            # then a smaller one, not a multiple of anything
    for o, d in ((o1, d1), (o2, d2)):
        t_a, f_a, t_b, f_b = both(o, d)
        assert (f_a >= 0).sum() > len(f_a) // 4
        assert np.array_equal(f_a, f_b)
        assert np.array_equal(t_a.view(np.uint32), t_b.view(np.uint32))
    # the second call's answers against the oracle: Flyscene::getIntersections restated, every triangle per ray
    from oracle import oracle_api as oa
    oa.build()
    osc = oa.Scene(verts, idx, None, [oa.make_mtl()])
    h = osc.closest(o2.cpu().numpy(), d2.cpu().numpy(), threads=8)
    assert np.array_equal(f_b, h["face"]), "face ids differ from the oracle's brute force"
    assert np.array_equal(t_b.view(np.uint32), h["t"].view(np.uint32)), "hit distances differ from the oracle's brute force"
    # ... and the forced group-by-group scheduler takes the same scratch layout
    s.set_flags(api.FLAG_SCHED_GROUPS)
    t_a, f_a, t_b, f_b = both(o2, d2)
    assert np.array_equal(f_a, f_b) and np.array_equal(t_a.view(np.uint32), t_b.view(np.uint32))
    s.set_flags(0)
    # sizes beyond the fixed array are refused, loudly, before anything runs
    o3, d3 = rays(api.NOALLOC_MAX_RAYS + 1)
    out_t = torch.zeros(api.NOALLOC_MAX_RAYS + 1, dtype=torch.float32, device=dev)
    out_f = torch.zeros(api.NOALLOC_MAX_RAYS + 1, dtype=torch.int32, device=dev)
    with pytest.raises(api.RtxError, match="at most 65536 rays"):
        s.trace_closest_noalloc(o3.data_ptr(), d3.data_ptr(), api.NOALLOC_MAX_RAYS + 1, out_t.data_ptr(), out_f.data_ptr())
    s.close()

import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
os.environ["RTX_LIB"] = "/root/repo/scratch/librtx_dbg.so"
import torch
import bench
from raytracing_project_amd import api, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "soup1M_4k_primary_shadow"
wl = bench.build_workload(name, api, scenes)
W, H = wl["W"], wl["H"]
scene = api.Scene(wl["verts"], wl["idx"], wl["tri_mat"], [api.make_material(**m) for m in wl["mats"]])
scene.set_lights(wl["lights"]); scene.commit(0)
opts = api.Options(*wl["opts"]); cam = api.make_camera(wl["view"], W, H)
rgb = torch.zeros((W * H, 3), dtype=torch.float32, device="cuda")
lib = api.lib()
lib.rtx_debug_fetch_times.argtypes = [C.c_void_p, C.c_uint, C.c_int]
fbuf = np.zeros(8192, np.uint64)
for it in range(3):
    if it == 2:
        torch.cuda.synchronize(); lib.rtx_debug_fetch_times(fbuf.ctypes.data, 8192, 1)
    scene.render_dev(cam, opts, (0, 0, W, H), rgb.data_ptr(), None, None)
st = scene.stats()
n = 8192
buf = np.zeros(2 * n, np.uint64)
lib = api.lib()
lib.rtx_debug_wave_times.argtypes = [C.c_void_p, C.c_uint]
rc = lib.rtx_debug_wave_times(buf.ctypes.data, 2 * n)
t0 = buf[0::2].astype(np.int64); t1 = buf[1::2].astype(np.int64)
ok = t1 > 0
t0, t1 = t0[ok], t1[ok]
start = t0.min()
dur = (t1 - t0) / 100.0   # us
end = (t1 - start) / 100.0
print("sched", os.environ.get("RTX_SCHED"), "rc", rc, "waves", ok.sum(), "cam ms", st.ms_trace_camera)
print("wave busy time us: mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % (dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
print("wave end time us:  mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f  max %.1f" % (end.mean(), *np.percentile(end, [10, 50, 90]), end.max()))
lib.rtx_debug_fetch_times(fbuf.ctypes.data, 8192, 0)
print("fetch time per wave us (all trace launches of the frame): mean %.1f max %.1f" % (fbuf.mean() / 100.0, fbuf.max() / 100.0))
print("mean busy / max end = %.3f" % (dur.mean() / end.max()))

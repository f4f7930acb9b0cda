"""Where Flyscene::initialize's time goes on the bunny (69k triangles): the C-ABI calls behind it, one by one."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_project_amd import api
import ctypes as C
z = np.load(os.path.join(ROOT, "tests", "golden", "scene_bunny.npz"))
t = [time.perf_counter()]
def lap(what):
    t.append(time.perf_counter()); print("%-52s %8.1f ms" % (what, (t[-1] - t[-2]) * 1e3), flush=True)
api.lib(); lap("load librtx.so")
n = api.lib().rtx_device_count(); lap("rtx_device_count (first HIP call)")
s = api.Scene(z["verts"], z["idx"], z["tri_mat"], [api.make_material(ka=(0, 0, 0), kd=(.7, .7, .7), ks=(.2, .2, .2), ns=10, ni=1, d=1, illum=2)]); lap("rtx_scene_create")
s.set_lights([[-1.0, 1.0, 1.0]]); lap("set_lights")
s.build(); lap("rtx_scene_build (host binned SAH)")
s.commit(0); lap("rtx_scene_commit (context, streams, upload)")
nl = api.lib().rtx_scene_build_reference_leafboxes(s.h, 0, 0); lap("rtx_scene_build_reference_leafboxes -> %d" % nl)
s.commit(0); lap("second commit (leaf list upload)")
import torch
rgb = torch.zeros((3840 * 2160, 3), dtype=torch.float32, device="cuda"); torch.cuda.synchronize(); lap("torch buffer")
cam = api.make_camera(np.array([[1, 0, 0, 0], [0, 1, 0, -0.1], [0, 0, 1, -0.4]], np.float32), 3840, 2160)
for k in range(3):
    s.render_dev(cam, api.Options(0, 1, 0, 0, 1, 0), (0, 0, 3840, 2160), rgb.data_ptr(), None, None); lap("rtx_render_dev 4K, call %d" % k)
host = np.empty((3840 * 2160, 3), np.float32)
for k in range(3):
    s.render(cam, api.Options(0, 1, 0, 0, 1, 0), (0, 0, 3840, 2160), True, False); lap("rtx_render 4K into host memory, call %d" % k)
# the same call into page-locked memory (rtx_host_alloc)
L = api.lib()
p = C.c_void_p()
assert L.rtx_host_alloc(3840 * 2160 * 12, C.byref(p)) == 0
opts = api.Options(0, 1, 0, 0, 1, 0)
for k in range(3):
    rc = L.rtx_render(s.h, C.byref(cam), C.byref(opts), api.Rect(0, 0, 3840, 2160), C.cast(p, C.POINTER(C.c_float)), None, None)
    assert rc == 0
    lap("rtx_render 4K into page-locked memory, call %d" % k)
L.rtx_host_free(p)

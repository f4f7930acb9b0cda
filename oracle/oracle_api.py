"""ctypes doors onto the CPU checker -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product path (raytracing_project_amd) never does.

  liboracle.so     plain-C restatement (oracle/rt_oracle.c)
  _ref/libref.so   the reference's own Intersect.cpp / Ray.h / camera.hpp, built in place
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_f = C.POINTER(C.c_float)


class Mtl(C.Structure):
    _fields_ = [("ka", C.c_float * 3), ("kd", C.c_float * 3), ("ks", C.c_float * 3),
                ("ns", C.c_float), ("ni", C.c_float), ("d", C.c_float), ("illum", C.c_int)]


class Hit(C.Structure):
    _fields_ = [("t", C.c_float), ("face", C.c_int), ("n", C.c_float * 3), ("p", C.c_float * 3),
                ("mat", C.c_int)]


HIT_DTYPE = np.dtype([("t", "<f4"), ("face", "<i4"), ("n", "<f4", 3), ("p", "<f4", 3), ("mat", "<i4")])


class Camera(C.Structure):
    _fields_ = [("view", C.c_float * 12), ("vp", C.c_float * 4), ("fovy", C.c_float),
                ("aspect", C.c_float), ("sub", C.c_float * 2)]


class Options(C.Structure):
    _fields_ = [("max_bounces", C.c_int), ("reflection", C.c_int), ("refraction", C.c_int),
                ("gi_rays", C.c_int), ("shadows", C.c_int), ("skip_zero_weight", C.c_int)]


class Mesh(C.Structure):
    _fields_ = [("nverts", C.c_int), ("ntris", C.c_int), ("nmats", C.c_int),
                ("verts", _f), ("idx", C.POINTER(C.c_uint32)), ("tri_mat", C.POINTER(C.c_int32)),
                ("mats", C.POINTER(Mtl)), ("mat_names", C.POINTER(C.c_char * 64))]


def build(force=False):
    """Compile the checker (gcc) and, where /root/reference exists, oracle/_ref."""
    if force or not os.path.exists(os.path.join(HERE, "liboracle.so")) or os.path.isdir("/root/reference"):
        subprocess.run(["make", "-s", "-C", HERE], check=True)


_lib = None
_ref = None


def fp(a):
    return a.ctypes.data_as(_f)


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.ora_scene_create.restype = C.c_void_p
        L.ora_scene_create.argtypes = [_f, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.c_int,
                                       C.POINTER(Mtl), C.c_int]
        for name in ("ora_scene_destroy", "ora_scene_set_spheres", "ora_scene_set_lights",
                     "ora_scene_set_options", "ora_scene_root_box", "ora_scene_face_normal",
                     "ora_closest", "ora_visible", "ora_trace", "ora_render", "ora_ray_counts",
                     "ora_closest_mt"):
            getattr(L, name).restype = None
        L.ora_scene_destroy.argtypes = [C.c_void_p]
        L.ora_scene_set_spheres.argtypes = [C.c_void_p, _f, C.POINTER(C.c_int32), C.c_int]
        L.ora_scene_set_lights.argtypes = [C.c_void_p, _f, C.c_int, _f, C.c_int]
        L.ora_scene_set_options.argtypes = [C.c_void_p, C.POINTER(Options)]
        L.ora_scene_add_material.argtypes = [C.c_void_p, C.POINTER(Mtl)]
        L.ora_scene_add_material.restype = C.c_int
        L.ora_scene_build_leafboxes.argtypes = [C.c_void_p, C.c_int, C.c_long]
        L.ora_scene_build_leafboxes.restype = C.c_int
        L.ora_scene_num_leaves.argtypes = [C.c_void_p]
        L.ora_scene_num_leaves.restype = C.c_int
        L.ora_scene_leaf.argtypes = [C.c_void_p, C.c_int, _f, _f, C.POINTER(C.c_int32), C.c_int]
        L.ora_scene_leaf.restype = C.c_int
        L.ora_scene_root_box.argtypes = [C.c_void_p, _f, _f]
        L.ora_scene_face_normal.argtypes = [C.c_void_p, C.c_int, _f, _f]
        L.ora_closest.argtypes = [C.c_void_p, _f, _f, C.c_int, C.c_void_p]
        L.ora_closest_mt.argtypes = [C.c_void_p, _f, _f, C.c_int, C.c_int, C.c_void_p]
        L.ora_visible.argtypes = [C.c_void_p, _f, _f, C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_int32),
                                  _f, C.POINTER(C.c_int32)]
        L.ora_trace.argtypes = [C.c_void_p, _f, _f, C.c_int, _f]
        L.ora_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                 _f, _f, C.POINTER(C.c_int32)]
        L.ora_ray_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.ora_debug_hit.argtypes = [C.c_void_p, _f, _f, C.c_int, C.c_void_p, C.POINTER(C.c_uint8)]
        L.ora_debug_hit.restype = None
        L.ora_debug_ray.argtypes = [C.c_void_p, _f, _f, C.c_int, _f, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_uint8), _f, C.c_int,
                                    C.POINTER(C.c_int)]
        L.ora_debug_ray.restype = C.c_int
        L.ora_box.restype = C.c_float
        L.ora_dot.restype = C.c_float
        L.ora_norm.restype = C.c_float
        L.ora_obj_load.argtypes = [C.c_char_p, C.POINTER(Mesh)]
        L.ora_obj_load.restype = C.c_int
        L.ora_mesh_free.argtypes = [C.POINTER(Mesh)]
        L.ora_ppm_write.argtypes = [C.c_char_p, _f, C.c_int, C.c_int, C.c_int]
        L.ora_ppm_write.restype = C.c_int
        L.ora_camera_point.argtypes = [C.POINTER(Camera), C.c_float, C.c_float, _f, _f]
        L.ora_sphere.argtypes = [_f, _f, _f, C.c_float, _f]
        L.ora_refract.argtypes = [_f, _f, _f, _f, _f, C.c_float, C.c_float, _f, _f]
        L.ora_resend.argtypes = [_f, _f, _f, _f, C.c_int, _f]
        L.ora_points_on_circle.argtypes = [_f, C.c_float, _f, C.c_int, _f]
        L.ora_points_on_circle.restype = C.c_int
        L.ora_bvh_trace.argtypes = [C.POINTER(C.c_uint32), _f, C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, _f, _f, _f, _f,
                                    C.c_int, C.c_int, _f, C.POINTER(C.c_int32), C.POINTER(C.c_uint32),
                                    C.POINTER(C.c_uint32)]
        L.ora_bvh_trace.restype = None
        L.ora_camera_rays.argtypes = [C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int, _f, _f]
        L.ora_camera_rays.restype = None
        _lib = L
    return _lib


def bvh_trace(nodes8, tri16, tri_face, root_lo, root_hi, o, d1, threads=1, nwild=0, layout_counts=False):
    """CPU mirror of the GPU traversal over the BVH exported by librtx: (t, face, nodes_visited, tris_tested)
    and, with layout_counts, also (group_steps, tris_full): 64-byte group fetches and triangle tests past the plane stage."""
    nodes8 = np.ascontiguousarray(nodes8, np.uint32)
    tri16 = np.ascontiguousarray(tri16, np.float32)
    tri_face = np.ascontiguousarray(tri_face, np.int32)
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d1 = np.ascontiguousarray(d1, np.float32).reshape(-1, 3)
    rl = np.ascontiguousarray(root_lo, np.float32)
    rh = np.ascontiguousarray(root_hi, np.float32)
    n = len(o)
    t = np.zeros(n, np.float32)
    face = np.zeros(n, np.int32)
    nv = np.zeros(n, np.uint32)
    tt = np.zeros(n, np.uint32)
    gs = np.zeros(n, np.uint32)
    tf = np.zeros(n, np.uint32)
    u32 = C.POINTER(C.c_uint32)
    lib().ora_bvh_trace2(nodes8.ctypes.data_as(u32), fp(tri16),
                         tri_face.ctypes.data_as(C.POINTER(C.c_int32)), len(tri_face), nwild, fp(rl), fp(rh), fp(o), fp(d1),
                         n, threads, fp(t), face.ctypes.data_as(C.POINTER(C.c_int32)),
                         nv.ctypes.data_as(u32), tt.ctypes.data_as(u32), gs.ctypes.data_as(u32), tf.ctypes.data_as(u32))
    if layout_counts:
        return t, face, nv, tt, gs, tf
    return t, face, nv, tt


def bvh_trace_shadow(nodes8, tri16, tri_face, root_lo, root_hi, o, d1, tmax, anyhit, threads=1, nwild=0):
    """Shadow-queue rays as the device walks them (bounded at tmax, optionally stopping at any blocker clearly before
    the light): (t, face, nodes_visited, tris_tested, group_steps, tris_full)."""
    nodes8 = np.ascontiguousarray(nodes8, np.uint32)
    tri16 = np.ascontiguousarray(tri16, np.float32)
    tri_face = np.ascontiguousarray(tri_face, np.int32)
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d1 = np.ascontiguousarray(d1, np.float32).reshape(-1, 3)
    tmax = np.ascontiguousarray(tmax, np.float32)
    rl = np.ascontiguousarray(root_lo, np.float32)
    rh = np.ascontiguousarray(root_hi, np.float32)
    n = len(o)
    t = np.zeros(n, np.float32)
    face = np.zeros(n, np.int32)
    nv, tt, gs, tf = (np.zeros(n, np.uint32) for _ in range(4))
    u32 = C.POINTER(C.c_uint32)
    L = lib()
    L.ora_bvh_trace_shadow.restype = None
    L.ora_bvh_trace_shadow(nodes8.ctypes.data_as(u32), fp(tri16), tri_face.ctypes.data_as(C.POINTER(C.c_int32)),
                           C.c_uint32(len(tri_face)), C.c_uint32(nwild), fp(rl), fp(rh), fp(o), fp(d1), fp(tmax), C.c_int(1 if anyhit else 0),
                           C.c_int(n), C.c_int(threads), fp(t), face.ctypes.data_as(C.POINTER(C.c_int32)), nv.ctypes.data_as(u32),
                           tt.ctypes.data_as(u32), gs.ctypes.data_as(u32), tf.ctypes.data_as(u32))
    return t, face, nv, tt, gs, tf


def camera_rays(cam, rect):
    x0, y0, x1, y1 = rect
    n = (x1 - x0) * (y1 - y0)
    o = np.zeros((n, 3), np.float32)
    d1 = np.zeros((n, 3), np.float32)
    lib().ora_camera_rays(C.byref(cam), x0, y0, x1, y1, fp(o), fp(d1))
    return o, d1


def ref():
    """The reference-built library, or None when it was never built (no /root/reference)."""
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libref.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference"):
                build(force=True)
            else:
                return None
        R = C.CDLL(path)
        R.ref_box.restype = C.c_float
        R.ref_dot.restype = C.c_float
        R.ref_norm.restype = C.c_float
        R.ref_sphere.argtypes = [_f, _f, _f, C.c_float, _f]
        R.ref_refract.argtypes = [_f, _f, _f, _f, _f, C.c_float, C.c_float, _f, _f]
        R.ref_resend.argtypes = [_f, _f, _f, _f, C.c_int, _f]
        R.ref_camera.argtypes = [_f, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _f, _f]
        if hasattr(R, "ref_load_mtl"):      # (a libref.so built before these doors existed lacks them)
            R.ref_load_mtl.argtypes = [C.c_char_p, _f, C.c_char_p, C.c_int]
            R.ref_load_mtl.restype = C.c_int
            R.ref_write_ppm.argtypes = [C.c_char_p, _f, C.c_int, C.c_int]
            R.ref_write_ppm.restype = None
            R.ref_default_mtl.argtypes = [_f]
            R.ref_default_mtl.restype = None
        _ref = R
    return _ref


def ref_load_mtl(path, cap=64):
    """Tucano::MaterialImporter::loadMTL run by the reference build: (names, float32 [n, 13] = ka kd ks ns ni d illum)."""
    R = ref()
    vals = np.zeros((cap, 13), np.float32)
    names = C.create_string_buffer(64 * cap)
    n = R.ref_load_mtl(str(path).encode(), fp(vals), names, cap)
    if n < 0:
        return None, None
    assert n <= cap
    return [names.raw[64 * i: 64 * i + 64].split(b"\0")[0].decode("latin-1") for i in range(n)], vals[:n].copy()


def ref_default_mtl():
    v = np.zeros(13, np.float32)
    ref().ref_default_mtl(fp(v))
    return v


def ref_write_ppm(path, frame):
    """Tucano::ImageImporter::writePPMImage run by the reference build on a SQUARE row-major frame [y][x][3], handed
    over the way raytraceScene does: pixel (x, y) at data[x][y] (src/flyscene.cpp:465)."""
    frame = np.ascontiguousarray(frame, np.float32)
    h, w, _ = frame.shape
    assert h == w, "the reference's writer is only defined for square frames (SURVEY F8)"
    data = np.ascontiguousarray(frame.transpose(1, 0, 2))      # data[x][y]
    ref().ref_write_ppm(str(path).encode(), fp(data), w, h)


def make_mtl(ka=(0.3, 0.3, 0.3), kd=(0.5, 0.5, 0.5), ks=(1.0, 1.0, 1.0), ns=10.0, ni=0.0, d=1.0, illum=0):
    m = Mtl()
    m.ka[:] = ka
    m.kd[:] = kd
    m.ks[:] = ks
    m.ns, m.ni, m.d, m.illum = ns, ni, d, illum
    return m


def make_camera(view34, width, height, fovy=60.0, aspect=None, sub=(0.0, 0.0)):
    c = Camera()
    c.view[:] = [float(x) for x in np.asarray(view34, np.float32).reshape(12)]
    c.vp[:] = [0.0, 0.0, float(width), float(height)]
    c.fovy = fovy
    c.aspect = np.float32(width) / np.float32(height) if aspect is None else aspect
    c.sub[:] = sub
    return c


class Scene:
    """A scene held by the C checker."""

    def __init__(self, verts, idx, tri_mat, mats):
        L = lib()
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        self.idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
        n = len(self.idx)
        self.tri_mat = (np.full(n, -1, np.int32) if tri_mat is None
                        else np.ascontiguousarray(tri_mat, np.int32))
        arr = (Mtl * max(1, len(mats)))(*mats)
        self.h = L.ora_scene_create(fp(self.verts), len(self.verts),
                                    self.idx.ctypes.data_as(C.POINTER(C.c_uint32)),
                                    self.tri_mat.ctypes.data_as(C.POINTER(C.c_int32)), n, arr, len(mats))
        self.nmats = len(mats)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ora_scene_destroy(self.h)
            self.h = None

    def set_spheres(self, center_radius4, mats):
        cr = np.ascontiguousarray(center_radius4, np.float32).reshape(-1, 4)
        m = np.ascontiguousarray(mats, np.int32)
        lib().ora_scene_set_spheres(self.h, fp(cr), m.ctypes.data_as(C.POINTER(C.c_int32)), len(cr))

    def add_material(self, m):
        return lib().ora_scene_add_material(self.h, C.byref(m))

    def set_lights(self, points, sphere_lights=None):
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        s = np.zeros((0, 4), np.float32) if sphere_lights is None else \
            np.ascontiguousarray(sphere_lights, np.float32).reshape(-1, 4)
        lib().ora_scene_set_lights(self.h, fp(p), len(p), fp(s), len(s))

    def set_options(self, max_bounces=2, reflection=1, refraction=1, gi_rays=8, shadows=1, skip_zero_weight=0):
        o = Options(max_bounces, reflection, refraction, gi_rays, shadows, skip_zero_weight)
        lib().ora_scene_set_options(self.h, C.byref(o))

    def build_leafboxes(self, max_faces=None, iter_cap=None):
        n = len(self.idx)
        if max_faces is None:
            max_faces = int(np.sqrt(float(n + 100)))
        if iter_cap is None:
            iter_cap = 64 * n + 4096
        return lib().ora_scene_build_leafboxes(self.h, max_faces, iter_cap)

    def leaves(self):
        L = lib()
        out = []
        for i in range(L.ora_scene_num_leaves(self.h)):
            bmin = np.zeros(3, np.float32)
            bmax = np.zeros(3, np.float32)
            faces = np.zeros(len(self.idx), np.int32)
            k = L.ora_scene_leaf(self.h, i, fp(bmin), fp(bmax), faces.ctypes.data_as(C.POINTER(C.c_int32)),
                                 len(faces))
            out.append((bmin, bmax, faces[:k].copy()))
        return out

    def root_box(self):
        a = np.zeros(3, np.float32)
        b = np.zeros(3, np.float32)
        lib().ora_scene_root_box(self.h, fp(a), fp(b))
        return a, b

    def closest(self, o, d, threads=1):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros(len(o), HIT_DTYPE)
        lib().ora_closest_mt(self.h, fp(o), fp(d), len(o), threads, out.ctypes.data_as(C.c_void_p))
        return out

    def num_leaves(self):
        return int(lib().ora_scene_num_leaves(self.h))

    def debug_hit(self, o, d, want_flags=False):
        """Hit search of Flyscene::ReflectDebugRay: HIT_DTYPE array (+ [n, leaves] flags)."""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros(len(o), HIT_DTYPE)
        flags = np.zeros((len(o), max(1, self.num_leaves())), np.uint8)
        lib().ora_debug_hit(self.h, fp(o), fp(d), len(o), out.ctypes.data_as(C.c_void_p),
                            flags.ctypes.data_as(C.POINTER(C.c_uint8)) if want_flags else None)
        return (out, flags[:, :self.num_leaves()]) if want_flags else out

    def debug_ray(self, origin, direction, jumps, max_segs=64):
        """Flyscene::ReflectDebugRay as data: (segments [k, 8], faces [k], box flags [leaves], light segments [m, 7])."""
        origin = np.ascontiguousarray(origin, np.float32); direction = np.ascontiguousarray(direction, np.float32)
        segs = np.zeros((max_segs, 8), np.float32); faces = np.zeros(max_segs, np.int32)
        flags = np.zeros(max(1, self.num_leaves()), np.uint8)
        ls = np.zeros((max_segs * 8, 7), np.float32); nls = C.c_int(0)
        k = lib().ora_debug_ray(self.h, fp(origin), fp(direction), int(jumps), fp(segs), faces.ctypes.data_as(C.POINTER(C.c_int32)),
                                max_segs, flags.ctypes.data_as(C.POINTER(C.c_uint8)), fp(ls), len(ls), C.byref(nls))
        return segs[:k], faces[:k], flags[:self.num_leaves()], ls[:nls.value]

    def trace_gi(self, o, d):
        """Flyscene::trace_global_illum(Ray(o, d)): [n, 4] = (Kd, t), zeros where nothing is hit."""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros((len(o), 4), np.float32)
        L = lib()
        L.ora_trace_gi.restype = None
        L.ora_trace_gi.argtypes = [C.c_void_p, _f, _f, C.c_int, _f]
        L.ora_trace_gi(self.h, fp(o), fp(d), len(o), fp(out))
        return out

    def visible(self, light, pos):
        light = np.ascontiguousarray(light, np.float32).reshape(-1, 3)
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        n = len(pos)
        vis = np.zeros(n, np.uint8)
        bm = np.zeros(n, np.int32)
        bt = np.zeros(n, np.float32)
        bf = np.zeros(n, np.int32)
        lib().ora_visible(self.h, fp(light), fp(pos), n, vis.ctypes.data_as(C.POINTER(C.c_uint8)),
                          bm.ctypes.data_as(C.POINTER(C.c_int32)), fp(bt),
                          bf.ctypes.data_as(C.POINTER(C.c_int32)))
        return vis, bm, bt, bf

    def trace(self, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        rgb = np.zeros((len(o), 3), np.float32)
        lib().ora_trace(self.h, fp(o), fp(d), len(o), fp(rgb))
        return rgb

    def render(self, cam, rect, threads=1, want_rgb=True, want_hits=True):
        x0, y0, x1, y1 = rect
        W, H = x1 - x0, y1 - y0
        rgb = np.zeros((H, W, 3), np.float32) if want_rgb else None
        t = np.zeros((H, W), np.float32) if want_hits else None
        face = np.zeros((H, W), np.int32) if want_hits else None
        lib().ora_render(self.h, C.byref(cam), x0, y0, x1, y1, threads,
                         fp(rgb) if want_rgb else None, fp(t) if want_hits else None,
                         face.ctypes.data_as(C.POINTER(C.c_int32)) if want_hits else None)
        return rgb, t, face

    def ray_counts(self):
        c = (C.c_uint64 * 5)()
        lib().ora_ray_counts(self.h, c)
        return dict(zip(("primary", "shadow", "reflection", "refraction", "gi"), [int(x) for x in c]))


def obj_load(path):
    m = Mesh()
    rc = lib().ora_obj_load(path.encode(), C.byref(m))
    if rc != 0:
        raise IOError("ora_obj_load(%s) -> %d" % (path, rc))
    verts = np.ctypeslib.as_array(m.verts, (m.nverts, 3)).copy() if m.nverts else np.zeros((0, 3), np.float32)
    idx = np.ctypeslib.as_array(m.idx, (m.ntris, 3)).copy() if m.ntris else np.zeros((0, 3), np.uint32)
    tm = np.ctypeslib.as_array(m.tri_mat, (m.ntris,)).copy() if m.ntris else np.zeros((0,), np.int32)
    mats = []
    names = []
    for i in range(m.nmats):
        src = m.mats[i]
        mats.append(make_mtl(tuple(src.ka), tuple(src.kd), tuple(src.ks), src.ns, src.ni, src.d, src.illum))
        names.append(m.mat_names[i].value.decode())
    lib().ora_mesh_free(C.byref(m))
    return verts, idx, tm, mats, names

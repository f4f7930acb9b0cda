"""ctypes binding of include/rtx.h.  Fails loudly when librtx.so is missing."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_LIB", os.path.join(HERE, "librtx.so"))   # RTX_LIB: A/B builds of the same library

_f = C.POINTER(C.c_float)
_i = C.POINTER(C.c_int32)
_u = C.POINTER(C.c_uint32)


class Material(C.Structure):
    _fields_ = [("ka", C.c_float * 3), ("kd", C.c_float * 3), ("ks", C.c_float * 3),
                ("ns", C.c_float), ("ni", C.c_float), ("d", C.c_float), ("illum", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("view", C.c_float * 12), ("viewport", C.c_float * 4), ("fovy", C.c_float),
                ("aspect", C.c_float), ("subpixel", C.c_float * 2)]


class Options(C.Structure):
    _fields_ = [("max_bounces", C.c_int32), ("reflection", C.c_int32), ("refraction", C.c_int32),
                ("gi_rays", C.c_int32), ("shadows", C.c_int32), ("skip_zero_weight", C.c_int32)]


class Rect(C.Structure):
    _fields_ = [("x0", C.c_int32), ("y0", C.c_int32), ("x1", C.c_int32), ("y1", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_reflection", C.c_uint64),
                ("rays_refraction", C.c_uint64), ("rays_gi", C.c_uint64), ("ms_trace", C.c_float),
                ("ms_total", C.c_float), ("launches_trace", C.c_uint32), ("ms_trace_camera", C.c_float),
                ("rays_out_of_domain", C.c_uint32), ("verify_checked", C.c_uint32), ("verify_mismatches", C.c_uint32)]


class SceneInfo(C.Structure):
    _fields_ = [("ntris", C.c_uint32), ("nverts", C.c_uint32), ("nnodes", C.c_uint32), ("nleaves", C.c_uint32),
                ("max_depth", C.c_uint32), ("nspheres", C.c_uint32), ("nlights", C.c_uint32),
                ("nmaterials", C.c_uint32), ("device_bytes", C.c_uint64), ("build_ms", C.c_float),
                ("nwild", C.c_uint32), ("nslots", C.c_uint32), ("ntop_groups", C.c_uint32),
                ("build_kind", C.c_uint32), ("device_build_ms", C.c_float), ("refit_ms", C.c_float), ("refits", C.c_uint32)]


# every symbol include/rtx.h declares
SYMBOLS = ["rtx_last_error", "rtx_device_count", "rtx_host_alloc", "rtx_host_free", "rtx_scene_create", "rtx_scene_destroy",
           "rtx_scene_add_material", "rtx_scene_set_spheres", "rtx_scene_set_lights", "rtx_scene_build", "rtx_scene_commit",
           "rtx_scene_get_tris",
           "rtx_scene_get_info", "rtx_scene_get_bvh", "rtx_trace_closest", "rtx_trace_closest_dev",
           "rtx_trace_visible", "rtx_render", "rtx_render_dev", "rtx_render_dev_enqueue", "rtx_render_finish", "rtx_tiles_count", "rtx_render_tiles_dev", "rtx_render_tiles_dev_enqueue", "rtx_render_tiles_batch_dev", "rtx_render_tiles_batch_dev_enqueue", "rtx_tiles_assemble_dev", "rtx_tiles_assemble_bytes_dev", "rtx_rgb_to_ppm8_dev",
           "rtx_get_stats", "rtx_scene_build_reference_leafboxes", "rtx_scene_set_leafboxes",
           "rtx_scene_get_leafboxes", "rtx_intersect_box", "rtx_trace_shade", "rtx_camera_points", "rtx_scene_set_flags",
           "rtx_scene_get_packed_bvh", "rtx_debug_hit", "rtx_trace_closest_noalloc", "rtx_scene_clone", "rtx_render_multi",
           "rtx_render_multi_dev", "rtx_intersect_primitive", "rtx_trace_gi", "rtx_scene_update_vertices", "rtx_scene_update_vertices_dev"]

MAX_BATCH = 64
NOALLOC_MAX_RAYS = 65536
FLAG_NO_ROOT_GATE, FLAG_SCHED_REFILL, FLAG_SCHED_GROUPS, FLAG_SCHED_SPLIT, FLAG_VERIFY, FLAG_FAST_TRIANGLES, FLAG_PARENT_LIST = 1, 2, 4, 8, 16, 32, 64
FLAG_DEVICE_BUILD = 128
UPDATE_REFIT, UPDATE_REBUILD = 0, 1

_lib = None


class RtxError(RuntimeError):
    pass


def _load_torch_first():
    """The torch wheel brings its own copy of the HIP / HSA runtime libraries, librtx.so binds to the
    ones under /opt/rocm.  A process can hold both when torch's are loaded first (bench.py, the tests);
    the other way round torch finds no device afterwards ("No HIP GPUs are available", seen on the
    MI355X box).  So where torch is installed it is imported before librtx.so is opened; without torch
    nothing happens here -- the library itself does not need it."""
    import importlib.util
    import sys
    if "torch" in sys.modules or importlib.util.find_spec("torch") is None:
        return
    try:
        import torch  # noqa: F401
    except Exception:   # a broken torch install must not keep the library from loading
        pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("librtx.so is missing (%s): build it with __graft_entry__.build(); "
                              "there is no CPU fallback" % LIB_PATH)
        _load_torch_first()
        L = C.CDLL(LIB_PATH)
        L.rtx_last_error.restype = C.c_char_p
        L.rtx_scene_create.argtypes = [_f, C.c_uint32, _u, _i, C.c_uint32, C.POINTER(Material), C.c_uint32,
                                       C.POINTER(C.c_void_p)]
        L.rtx_scene_destroy.argtypes = [C.c_void_p]
        L.rtx_host_alloc.argtypes = [C.c_uint64, C.POINTER(C.c_void_p)]
        L.rtx_host_free.argtypes = [C.c_void_p]
        L.rtx_scene_add_material.argtypes = [C.c_void_p, C.POINTER(Material)]
        L.rtx_scene_set_spheres.argtypes = [C.c_void_p, _f, _i, C.c_uint32]
        L.rtx_scene_set_lights.argtypes = [C.c_void_p, _f, C.c_uint32, _f, C.c_uint32]
        L.rtx_scene_commit.argtypes = [C.c_void_p, C.c_int]
        L.rtx_scene_build.argtypes = [C.c_void_p]
        L.rtx_scene_get_tris.argtypes = [C.c_void_p, _f]
        L.rtx_scene_get_info.argtypes = [C.c_void_p, C.POINTER(SceneInfo)]
        L.rtx_scene_get_bvh.argtypes = [C.c_void_p, _u, _u]
        L.rtx_trace_closest.argtypes = [C.c_void_p, _f, _f, C.c_uint32, _f, _i, _f, _f, _i]
        L.rtx_trace_closest_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rtx_intersect_primitive.argtypes = [C.c_int, C.c_int32, _f, _f, _f, C.c_uint32, _f, _i, _f, _f]
        L.rtx_trace_gi.argtypes = [C.c_void_p, _f, _f, C.c_uint32, _f]
        L.rtx_scene_clone.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.rtx_render_multi.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(Camera), C.POINTER(Options), Rect, C.c_int32, _f, _f, _i]
        L.rtx_render_multi_dev.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.POINTER(Camera), C.POINTER(Options), Rect, C.c_int32,
                                           C.c_void_p, C.c_void_p, C.c_void_p]
        L.rtx_trace_closest_noalloc.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rtx_trace_visible.argtypes = [C.c_void_p, _f, _f, C.c_uint32, C.POINTER(C.c_uint8), _i]
        L.rtx_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Options), Rect, _f, _f, _i]
        L.rtx_render_dev.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Options), Rect, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
        L.rtx_render_dev_enqueue.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Options), Rect, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
        L.rtx_render_finish.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.rtx_get_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.rtx_scene_build_reference_leafboxes.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        L.rtx_scene_set_leafboxes.argtypes = [C.c_void_p, _f, _f, _u, _u, C.c_uint32, _i, C.c_uint32]
        L.rtx_scene_get_leafboxes.argtypes = [C.c_void_p, _f, _f, _u, _u, _i, _u, _u]
        L.rtx_intersect_box.argtypes = [C.c_void_p, _f, _f, _f, _f, C.c_uint32, _f]
        L.rtx_trace_shade.argtypes = [C.c_void_p, C.POINTER(Options), _f, _f, C.c_uint32, C.c_int32, C.c_int32, _f]
        L.rtx_camera_points.argtypes = [C.c_void_p, C.POINTER(Camera), _f, C.c_uint32, _f, _f]
        L.rtx_scene_set_flags.argtypes = [C.c_void_p, C.c_uint32]
        L.rtx_scene_update_vertices.argtypes = [C.c_void_p, _f, C.c_uint32, C.c_int32]
        L.rtx_scene_update_vertices_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32]
        L.rtx_scene_get_packed_bvh.argtypes = [C.c_void_p, _u]
        L.rtx_debug_hit.argtypes = [C.c_void_p, _f, _f, C.c_uint32, _f, C.POINTER(C.c_int32), _f, _f, C.POINTER(C.c_uint8)]
        L.rtx_tiles_count.argtypes = [Rect, C.c_int32, C.c_int32, C.c_int32]
        L.rtx_render_tiles_dev.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Options), Rect, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rtx_render_tiles_dev_enqueue.argtypes = [C.c_void_p, C.POINTER(Camera), C.POINTER(Options), Rect, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
        for fn in (L.rtx_render_tiles_batch_dev, L.rtx_render_tiles_batch_dev_enqueue):
            fn.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int32, C.POINTER(Options), Rect, C.c_int32, C.c_int32, C.c_int32,
                           C.c_void_p, C.c_void_p, C.c_void_p]
        L.rtx_tiles_assemble_dev.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, Rect, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.rtx_tiles_assemble_bytes_dev.argtypes = [C.c_void_p, C.c_int32, C.c_uint64, Rect, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.rtx_rgb_to_ppm8_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc < 0:
        raise RtxError("rtx error %d: %s" % (rc, lib().rtx_last_error().decode()))
    return rc


def make_material(ka=(0.3, 0.3, 0.3), kd=(0.5, 0.5, 0.5), ks=(1.0, 1.0, 1.0), ns=10.0, ni=0.0, d=1.0, illum=0):
    m = Material()
    m.ka[:] = ka
    m.kd[:] = kd
    m.ks[:] = ks
    m.ns, m.ni, m.d, m.illum = ns, ni, d, illum
    return m


def make_camera(view34, width, height, fovy=60.0, aspect=None, subpixel=(0.0, 0.0)):
    c = Camera()
    c.view[:] = [float(x) for x in np.asarray(view34, np.float32).reshape(12)]
    c.viewport[:] = [0.0, 0.0, float(width), float(height)]
    c.fovy = fovy
    c.aspect = float(np.float32(width) / np.float32(height)) if aspect is None else aspect
    c.subpixel[:] = subpixel
    return c


def fp(a):
    return a.ctypes.data_as(_f)


def tiles_count(rect, tile_size, part, nparts):
    x0, y0, x1, y1 = rect
    return _check(lib().rtx_tiles_count(Rect(x0, y0, x1, y1), tile_size, part, nparts))


class Scene:
    """Owner of one rtx_scene handle."""

    def __init__(self, verts, idx, tri_mat, mats):
        L = lib()
        self.verts = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        self.idx = np.ascontiguousarray(idx, np.uint32).reshape(-1, 3)
        n = len(self.idx)
        tm = None if tri_mat is None else np.ascontiguousarray(tri_mat, np.int32)
        arr = (Material * max(1, len(mats)))(*mats)
        h = C.c_void_p()
        _check(L.rtx_scene_create(fp(self.verts), len(self.verts), self.idx.ctypes.data_as(_u),
                                  None if tm is None else tm.ctypes.data_as(_i), n, arr, len(mats), C.byref(h)))
        self.h = h

    def clone(self):
        """rtx_scene_clone: a second handle on the same scene (host data + built BVH), to commit on another device."""
        h = C.c_void_p()
        _check(lib().rtx_scene_clone(self.h, C.byref(h)))
        other = Scene.__new__(Scene)
        other.verts, other.idx, other.h = self.verts, self.idx, h
        return other

    def close(self):
        if getattr(self, "h", None):
            lib().rtx_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def add_material(self, m):
        return _check(lib().rtx_scene_add_material(self.h, C.byref(m)))

    def set_spheres(self, center_radius4, mats):
        cr = np.ascontiguousarray(center_radius4, np.float32).reshape(-1, 4)
        m = np.ascontiguousarray(mats, np.int32)
        _check(lib().rtx_scene_set_spheres(self.h, fp(cr), m.ctypes.data_as(_i), len(cr)))

    def set_lights(self, points, sphere_lights=None):
        p = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
        s = np.zeros((0, 4), np.float32) if sphere_lights is None else \
            np.ascontiguousarray(sphere_lights, np.float32).reshape(-1, 4)
        _check(lib().rtx_scene_set_lights(self.h, fp(p), len(p), fp(s), len(s)))

    def set_flags(self, flags):
        """RTX_FLAG_* bits (include/rtx.h)."""
        _check(lib().rtx_scene_set_flags(self.h, flags))

    def build(self):
        _check(lib().rtx_scene_build(self.h))

    def tris(self):
        i = self.info()
        out = np.zeros((max(1, i.ntris), 16), np.float32)
        _check(lib().rtx_scene_get_tris(self.h, fp(out)))
        return out[:i.ntris]

    def build_reference_leafboxes(self, max_faces=0, iter_cap=0):
        return _check(lib().rtx_scene_build_reference_leafboxes(self.h, max_faces, iter_cap))

    def leafboxes(self):
        nl, nf = C.c_uint32(), C.c_uint32()
        _check(lib().rtx_scene_get_leafboxes(self.h, None, None, None, None, None, C.byref(nl), C.byref(nf)))
        lo = np.zeros((nl.value, 3), np.float32)
        hi = np.zeros((nl.value, 3), np.float32)
        st = np.zeros(nl.value, np.uint32)
        ct = np.zeros(nl.value, np.uint32)
        fa = np.zeros(max(1, nf.value), np.int32)
        _check(lib().rtx_scene_get_leafboxes(self.h, fp(lo), fp(hi), st.ctypes.data_as(_u), ct.ctypes.data_as(_u),
                                             fa.ctypes.data_as(_i), None, None))
        return [(lo[i], hi[i], fa[st[i]:st[i] + ct[i]].copy()) for i in range(nl.value)]

    def intersect_box(self, bmin, bmax, o, d):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        t = np.zeros(len(o), np.float32)
        _check(lib().rtx_intersect_box(self.h, fp(np.ascontiguousarray(bmin, np.float32)),
                                       fp(np.ascontiguousarray(bmax, np.float32)), fp(o), fp(d), len(o), fp(t)))
        return t

    def trace_shade(self, opts, o, d, bounce=0, inside=False):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        rgb = np.zeros((len(o), 3), np.float32)
        _check(lib().rtx_trace_shade(self.h, C.byref(opts), fp(o), fp(d), len(o), bounce, 1 if inside else 0, fp(rgb)))
        return rgb

    def camera_points(self, cam, ij):
        ij = np.ascontiguousarray(ij, np.float32).reshape(-1, 2)
        w = np.zeros((len(ij), 3), np.float32)
        c = np.zeros(3, np.float32)
        _check(lib().rtx_camera_points(self.h, C.byref(cam), fp(ij), len(ij), fp(w), fp(c)))
        return w, c

    def commit(self, device=0):
        _check(lib().rtx_scene_commit(self.h, device))

    def update_vertices(self, verts, how=UPDATE_REFIT):
        """rtx_scene_update_vertices (host array) / rtx_scene_update_vertices_dev (a torch tensor on the scene's device)."""
        if hasattr(verts, "data_ptr"):
            assert verts.is_cuda and verts.is_contiguous() and verts.numel() == self.verts.size
            _check(lib().rtx_scene_update_vertices_dev(self.h, verts.data_ptr(), len(self.verts), how))
            return
        v = np.ascontiguousarray(verts, np.float32).reshape(-1, 3)
        _check(lib().rtx_scene_update_vertices(self.h, fp(v), len(v), how))
        self.verts = v

    def info(self):
        i = SceneInfo()
        _check(lib().rtx_scene_get_info(self.h, C.byref(i)))
        return i

    def bvh(self):
        i = self.info()
        nodes = np.zeros((i.nnodes, 8), np.uint32)
        order = np.zeros(max(1, i.ntris), np.uint32)
        _check(lib().rtx_scene_get_bvh(self.h, nodes.ctypes.data_as(_u), order.ctypes.data_as(_u)))
        return nodes, order[:i.ntris]

    def debug_hit(self, o, d, want_flags=False):
        """rtx_debug_hit: (t, face, normal, point[, box_flags])."""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        t = np.zeros(n, np.float32); face = np.zeros(n, np.int32)
        nrm = np.zeros((n, 3), np.float32); pt = np.zeros((n, 3), np.float32)
        nl = C.c_uint32(); nf = C.c_uint32()
        _check(lib().rtx_scene_get_leafboxes(self.h, None, None, None, None, None, C.byref(nl), C.byref(nf)))
        flags = np.zeros((n, max(1, nl.value)), np.uint8) if want_flags else None
        _check(lib().rtx_debug_hit(self.h, fp(o), fp(d), n, fp(t), face.ctypes.data_as(C.POINTER(C.c_int32)), fp(nrm), fp(pt),
                                   flags.ctypes.data_as(C.POINTER(C.c_uint8)) if want_flags else None))
        return (t, face, nrm, pt, flags[:, :nl.value]) if want_flags else (t, face, nrm, pt)

    def packed_bvh(self):
        """[ngroups, 16] uint32: the 64-byte groups as the kernels load them."""
        i = self.info()
        g = np.zeros((max(1, i.nnodes // 4), 16), np.uint32)
        _check(lib().rtx_scene_get_packed_bvh(self.h, g.ctypes.data_as(_u)))
        return g[: i.nnodes // 4]

    def trace_closest(self, o, d, full=False):
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        n = len(o)
        t = np.zeros(n, np.float32)
        face = np.zeros(n, np.int32)
        if not full:
            _check(lib().rtx_trace_closest(self.h, fp(o), fp(d), n, fp(t), face.ctypes.data_as(_i), None, None, None))
            return t, face
        nrm = np.zeros((n, 3), np.float32)
        pt = np.zeros((n, 3), np.float32)
        mat = np.zeros(n, np.int32)
        _check(lib().rtx_trace_closest(self.h, fp(o), fp(d), n, fp(t), face.ctypes.data_as(_i), fp(nrm), fp(pt),
                                       mat.ctypes.data_as(_i)))
        return t, face, nrm, pt, mat

    def trace_closest_dev(self, o_ptr, d_ptr, n, t_ptr, face_ptr):
        _check(lib().rtx_trace_closest_dev(self.h, o_ptr, d_ptr, n, t_ptr, face_ptr))

    def trace_closest_noalloc(self, o_ptr, d_ptr, n, t_ptr, face_ptr):
        """rtx_trace_closest_noalloc: device pointers, at most NOALLOC_MAX_RAYS rays, no allocation inside."""
        _check(lib().rtx_trace_closest_noalloc(self.h, o_ptr, d_ptr, n, t_ptr, face_ptr))

    def trace_gi(self, o, d):
        """rtx_trace_gi: Flyscene::trace_global_illum(Ray(o, d)) -> [n, 4] (Kd, t)."""
        o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
        out = np.zeros((len(o), 4), np.float32)
        _check(lib().rtx_trace_gi(self.h, fp(o), fp(d), len(o), fp(out)))
        return out

    def trace_visible(self, light, pos):
        light = np.ascontiguousarray(light, np.float32).reshape(-1, 3)
        pos = np.ascontiguousarray(pos, np.float32).reshape(-1, 3)
        n = len(pos)
        vis = np.zeros(n, np.uint8)
        bm = np.zeros(n, np.int32)
        _check(lib().rtx_trace_visible(self.h, fp(light), fp(pos), n, vis.ctypes.data_as(C.POINTER(C.c_uint8)),
                                       bm.ctypes.data_as(_i)))
        return vis, bm

    def render(self, cam, opts, rect, want_rgb=True, want_hits=True):
        x0, y0, x1, y1 = rect
        W, H = x1 - x0, y1 - y0
        rgb = np.zeros((H, W, 3), np.float32) if want_rgb else None
        t = np.zeros((H, W), np.float32) if want_hits else None
        face = np.zeros((H, W), np.int32) if want_hits else None
        _check(lib().rtx_render(self.h, C.byref(cam), C.byref(opts), Rect(x0, y0, x1, y1),
                                fp(rgb) if want_rgb else None, fp(t) if want_hits else None,
                                face.ctypes.data_as(_i) if want_hits else None))
        return rgb, t, face

    def render_dev(self, cam, opts, rect, rgb_ptr, t_ptr, face_ptr):
        x0, y0, x1, y1 = rect
        _check(lib().rtx_render_dev(self.h, C.byref(cam), C.byref(opts), Rect(x0, y0, x1, y1), rgb_ptr, t_ptr,
                                    face_ptr))

    def render_dev_enqueue(self, cam, opts, rect, rgb_ptr, t_ptr, face_ptr):
        """render_dev without the wait; pair every call with render_finish() (two may be in flight)."""
        x0, y0, x1, y1 = rect
        _check(lib().rtx_render_dev_enqueue(self.h, C.byref(cam), C.byref(opts), Rect(x0, y0, x1, y1), rgb_ptr, t_ptr,
                                            face_ptr))

    def render_finish(self):
        """Waits for the oldest enqueued render call; returns its Stats."""
        s = Stats()
        _check(lib().rtx_render_finish(self.h, C.byref(s)))
        return s

    def render_tiles_dev(self, cam, opts, rect, tile_size, part, nparts, rgb_ptr, t_ptr, face_ptr):
        x0, y0, x1, y1 = rect
        _check(lib().rtx_render_tiles_dev(self.h, C.byref(cam), C.byref(opts), Rect(x0, y0, x1, y1), tile_size, part,
                                          nparts, rgb_ptr, t_ptr, face_ptr))

    def render_tiles_dev_enqueue(self, cam, opts, rect, tile_size, part, nparts, rgb_ptr, t_ptr, face_ptr):
        x0, y0, x1, y1 = rect
        _check(lib().rtx_render_tiles_dev_enqueue(self.h, C.byref(cam), C.byref(opts), Rect(x0, y0, x1, y1), tile_size, part,
                                                  nparts, rgb_ptr, t_ptr, face_ptr))

    def render_tiles_batch_dev(self, cams, opts, rect, tile_size, part, nparts, rgb_ptr, t_ptr, face_ptr, enqueue=False):
        """rtx_render_tiles_batch_dev(_enqueue): len(cams) frames in one set of launches; frame f's pixels follow frame f-1's."""
        x0, y0, x1, y1 = rect
        arr = (Camera * len(cams))(*cams)
        fn = lib().rtx_render_tiles_batch_dev_enqueue if enqueue else lib().rtx_render_tiles_batch_dev
        _check(fn(self.h, arr, len(cams), C.byref(opts), Rect(x0, y0, x1, y1), tile_size, part, nparts, rgb_ptr, t_ptr, face_ptr))

    def stats(self):
        s = Stats()
        _check(lib().rtx_get_stats(self.h, C.byref(s)))
        return s


def tiles_assemble_dev(parts_ptr, nparts, part_stride_pixels, rect, tile_size, words_per_pixel, out_ptr, stream=None):
    """rtx_tiles_assemble_dev: compact tile buffers of all parts -> row-major rectangle, on `stream` (no wait)."""
    x0, y0, x1, y1 = rect
    _check(lib().rtx_tiles_assemble_dev(parts_ptr, nparts, part_stride_pixels, Rect(x0, y0, x1, y1), tile_size,
                                        words_per_pixel, out_ptr, stream))


def tiles_assemble_bytes_dev(parts_ptr, nparts, part_stride_pixels, rect, tile_size, bytes_per_pixel, out_ptr, stream=None):
    """rtx_tiles_assemble_bytes_dev: the same for pixels of any size in bytes (3: the P3 bytes of rgb_to_ppm8_dev)."""
    x0, y0, x1, y1 = rect
    _check(lib().rtx_tiles_assemble_bytes_dev(parts_ptr, nparts, part_stride_pixels, Rect(x0, y0, x1, y1), tile_size,
                                              bytes_per_pixel, out_ptr, stream))


def rgb_to_ppm8_dev(rgb_ptr, npixels, out_ptr, stream=None):
    """rtx_rgb_to_ppm8_dev: min(255, (int)(255 c)) per channel (ppmIO.hpp:145) -> 3 bytes per pixel, on `stream` (no wait)."""
    _check(lib().rtx_rgb_to_ppm8_dev(rgb_ptr, npixels, out_ptr, stream))


def render_multi(scenes, cam, opts, rect, tile_size=32, want_rgb=True, want_hits=True):
    """rtx_render_multi: part i of the tile grid on scenes[i]'s device, the frame assembled on scenes[0]'s; host arrays back."""
    x0, y0, x1, y1 = rect
    W, H = x1 - x0, y1 - y0
    rgb = np.zeros((H, W, 3), np.float32) if want_rgb else None
    t = np.zeros((H, W), np.float32) if want_hits else None
    face = np.zeros((H, W), np.int32) if want_hits else None
    hs = (C.c_void_p * len(scenes))(*[s.h for s in scenes])
    _check(lib().rtx_render_multi(hs, len(scenes), C.byref(cam), C.byref(opts), Rect(x0, y0, x1, y1), tile_size,
                                  fp(rgb) if want_rgb else None, fp(t) if want_hits else None,
                                  face.ctypes.data_as(_i) if want_hits else None))
    return rgb, t, face


def render_multi_dev(scenes, cam, opts, rect, tile_size, rgb_ptr, t_ptr, face_ptr):
    x0, y0, x1, y1 = rect
    hs = (C.c_void_p * len(scenes))(*[s.h for s in scenes])
    _check(lib().rtx_render_multi_dev(hs, len(scenes), C.byref(cam), C.byref(opts), Rect(x0, y0, x1, y1), tile_size,
                                      rgb_ptr, t_ptr, face_ptr))


PRIM_TRIANGLE, PRIM_PLANE, PRIM_BOX, PRIM_SPHERE = 0, 1, 2, 3


def intersect_primitive(kind, prim, o, d, device=0):
    """rtx_intersect_primitive: Intersect::X::intersects(Ray&) for n Ray objects (o, d as the Ray holds them) -> (t, face, normal, point)."""
    prim = np.ascontiguousarray(prim, np.float32).reshape(-1)
    o = np.ascontiguousarray(o, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(d, np.float32).reshape(-1, 3)
    n = len(o)
    t = np.zeros(n, np.float32); face = np.zeros(n, np.int32)
    nrm = np.zeros((n, 3), np.float32); pt = np.zeros((n, 3), np.float32)
    _check(lib().rtx_intersect_primitive(device, kind, fp(prim), fp(o), fp(d), n, fp(t), face.ctypes.data_as(_i), fp(nrm), fp(pt)))
    return t, face, nrm, pt

"""Deterministic synthetic scenes and cameras for the BASELINE.json configs.

Everything is plain numpy (no RNG state outside splitmix64), so the CPU checker and the GPU
library are fed bit-identical inputs here and on the GPU box.
"""
import numpy as np

MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n consecutive splitmix64 outputs for `seed` (uint64 array)."""
    with np.errstate(over="ignore"):
        k = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + k * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _unit_floats(seed, n):
    """floats in [0,1): top 24 bits * 2^-24."""
    return ((splitmix64(seed, n) >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def soup(ntris, seed=0x5EED0001, h=0.02):
    """Random triangle soup (BASELINE config 4/5): centre ~ U[-1,1]^3, edges ~ U[-h,h]^3,
    v0 = c, v1 = c + e1, v2 = c + e2; 9 consecutive draws per triangle."""
    u = _unit_floats(seed, 9 * ntris).reshape(ntris, 9)
    c = (np.float32(2.0) * u[:, 0:3] - np.float32(1.0)).astype(np.float32)
    e1 = (np.float32(h) * (np.float32(2.0) * u[:, 3:6] - np.float32(1.0))).astype(np.float32)
    e2 = (np.float32(h) * (np.float32(2.0) * u[:, 6:9] - np.float32(1.0))).astype(np.float32)
    verts = np.empty((ntris, 3, 3), np.float32)
    verts[:, 0] = c
    verts[:, 1] = c + e1
    verts[:, 2] = c + e2
    idx = np.arange(3 * ntris, dtype=np.uint32).reshape(ntris, 3)
    return verts.reshape(-1, 3), idx


def blob(nu=187, nv=188):
    """Closed bumpy 'bunny-class' surface: a lat-long sphere (69,936 triangles at the defaults;
    the Stanford bunny of BASELINE config 2 has 69,451) whose radius is modulated by a few
    low-frequency lobes."""
    th = (np.arange(nu + 1, dtype=np.float64) / nu) * np.pi           # polar
    ph = (np.arange(nv, dtype=np.float64) / nv) * 2.0 * np.pi         # azimuth
    T, P = np.meshgrid(th, ph, indexing="ij")
    r = 0.55 + 0.12 * np.sin(3 * T) * np.cos(2 * P) + 0.08 * np.sin(5 * P + 1.0) * np.sin(T) ** 2 \
        + 0.15 * np.exp(-((T - 0.6) ** 2) / 0.05) * (np.cos(P - 0.8) > 0.7)
    x = r * np.sin(T) * np.cos(P)
    y = r * np.cos(T)
    z = r * np.sin(T) * np.sin(P)
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    i = np.arange(nu)[:, None]
    j = np.arange(nv)[None, :]
    a = (i * nv + j)
    b = (i * nv + (j + 1) % nv)
    c = ((i + 1) * nv + j)
    d = ((i + 1) * nv + (j + 1) % nv)
    # the pole rings collapse to a point: drop their zero-area triangles (a == b at the first
    # ring, c == d at the last), which leaves 2*nu*nv - 2*nv = 69,936 triangles at the defaults
    t1 = np.stack([a, c, b], -1)[1:].reshape(-1, 3)
    t2 = np.stack([b, c, d], -1)[:-1].reshape(-1, 3)
    idx = np.concatenate([t1, t2]).astype(np.uint32)
    return verts, idx


def hall(columns=32, seg=64, rings=64):
    """Closed room with a colonnade (BASELINE config 3 stand-in, 'Sponza-class'):
    columns*seg*rings*2 triangles of tessellated cylinders (262,144 at the defaults)
    plus the 12 triangles of the room box.  Returns verts, idx, tri_mat (0 = walls, 1 = columns)."""
    vs, ts, ms = [], [], []
    # room: x in [-4,4], y in [0,3], z in [-8,8]; normals do not matter to the tracer
    lo = np.array([-4.0, 0.0, -8.0])
    hi = np.array([4.0, 3.0, 8.0])
    corners = np.array([[lo[0], lo[1], lo[2]], [hi[0], lo[1], lo[2]], [hi[0], hi[1], lo[2]], [lo[0], hi[1], lo[2]],
                        [lo[0], lo[1], hi[2]], [hi[0], lo[1], hi[2]], [hi[0], hi[1], hi[2]], [lo[0], hi[1], hi[2]]])
    quads = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)]
    vs.append(corners)
    for q in quads:
        ts.append([q[0], q[1], q[2]])
        ts.append([q[0], q[2], q[3]])
        ms += [0, 0]
    base = 8
    per_side = columns // 2
    for c in range(columns):
        side = -1.0 if c < per_side else 1.0
        k = c % per_side
        cx = side * 2.2
        cz = -7.0 + 14.0 * (k + 0.5) / per_side
        ang = (np.arange(seg, dtype=np.float64) / seg) * 2 * np.pi
        yy = (np.arange(rings + 1, dtype=np.float64) / rings) * 3.0
        A, Y = np.meshgrid(ang, yy, indexing="xy")          # (rings+1, seg)
        rad = 0.28 + 0.05 * np.cos(8 * A) * (0.5 + 0.5 * np.cos(Y * 2.0))
        X = cx + rad * np.cos(A)
        Z = cz + rad * np.sin(A)
        v = np.stack([X, Y, Z], -1).reshape(-1, 3)
        vs.append(v)
        r = np.arange(rings)[:, None]
        s = np.arange(seg)[None, :]
        a = base + r * seg + s
        b = base + r * seg + (s + 1) % seg
        cc = base + (r + 1) * seg + s
        d = base + (r + 1) * seg + (s + 1) % seg
        ts.append(np.stack([a, b, cc], -1).reshape(-1, 3))
        ts.append(np.stack([b, d, cc], -1).reshape(-1, 3))
        ms += [1] * (2 * seg * rings)
        base += (rings + 1) * seg
    verts = np.concatenate(vs).astype(np.float32)
    idx = np.concatenate([np.asarray(t).reshape(-1, 3) for t in ts]).astype(np.uint32)
    return verts, idx, np.asarray(ms, np.int32)


def look_at_view(eye, target, up=(0.0, 1.0, 0.0)):
    """Row-major 3x4 view matrix [R|t] (world -> camera, camera looks down -z)."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, np.asarray(up, np.float64))
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    R = np.stack([s, u, -f])
    t = -R @ eye
    return np.concatenate([R, t[:, None]], 1).astype(np.float32)


def orbit_view(eye, target, k, step_deg=0.25):
    """Frame k of a camera path: `eye` turned about the vertical axis through `target` by k * step_deg degrees
    (frame 0 = look_at_view(eye, target)); what bench.py renders as consecutive frames of an animation."""
    eye = np.asarray(eye, np.float64)
    target = np.asarray(target, np.float64)
    a = np.deg2rad(step_deg * k)
    c, s = np.cos(a), np.sin(a)
    r = eye - target
    return look_at_view(target + np.array([c * r[0] + s * r[2], r[1], -s * r[0] + c * r[2]]), target)


def default_view():
    """Tucano::Flycamera default: identity rotation, translation (0,0,-2)
    (dependencies/tucano/tucano/utils/flycamera.hpp:76-86,166-191) -> centre (0,0,2)."""
    return np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -2]], np.float32)


# The 13 preview spheres of the reference (src/flyscene.hpp:177-206), positions * SCALE (0.5).
SPHERE_POS_W = np.array([
    [-0.5, 0.25, 0.8, 0.75], [0.75, 0.25, 0.7, 0.75], [0.6, 0.25, -0.4, 0.75], [-0.45, 0.25, -0.4, 0.75],
    [0.0, 0.167, 1.2, 0.5], [1.0, 0.167, 1.2, 0.5], [-1.0, 0.25, 1.2, 0.75], [0.0, 0.33, -2.2, 1.0],
    [0.0, 0.33, 2.2, 1.0], [-3.0, 0.167, -2.2, 0.5], [3.0, 0.25, -2.2, 0.75], [3.5, 0.33, 0.0, 1.0],
    [-3.5, 0.167, 0.0, 0.5]], np.float32) * np.float32(0.5)
SPHERE_COLORS = np.array([
    [0.8, 0.2401, 0.0], [0.5, 0.5, 0.0], [0.98824, 0.72941, 0.01176], [0.0, 0.5, 0.5], [1.0, 0.0, 1.0],
    [1.0, 0.75, 1.0], [0.5, 1.0, 0.75], [0.4, 0.3, 1.0], [0.8, 0.667, 0.0], [1.0, 0.75, 1.0], [0.0, 0.7, 1.0],
    [1.0, 0.5, 0.3], [0.0, 0.3, 0.123]], np.float32)


def reference_spheres():
    """(center_radius4, ka, kd) as Scene::addSphere stores them (src/Scene.h:36-49):
    radius = w / 3, Ka = color / 3, Kd = color, Ks = .5, Ns = 10, d = 0, Ni = 1, illum = 2."""
    cr = SPHERE_POS_W.copy()
    cr[:, 3] = SPHERE_POS_W[:, 3] / np.float32(3.0)
    ka = (SPHERE_COLORS / np.float32(3.0)).astype(np.float32)
    return cr, ka, SPHERE_COLORS.copy()

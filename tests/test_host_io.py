"""Host-side I/O of the C++ facade (raytracing_project_amd/host/flyscene.hpp) -- SURVEY.md 8(f)
ranks 1 and 2, the steps before and after the hot path: the OBJ + MTL loader with Tucano's face
order, the PLY reader, the P3 writer.  No GPU: tests/cpp/host_io_check.cpp is compiled here and
its output compared with the oracle's loader / writer (oracle/rt_oracle_io.c) and with a numpy
PLY reader kept in this file (test infrastructure)."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MODELS = "/root/reference/resources/models"


@pytest.fixture(scope="module")
def tool():
    import __graft_entry__ as g
    src = os.path.join(ROOT, "tests", "cpp", "host_io_check.cpp")
    exe = os.path.join(ROOT, "tests", "cpp", "host_io_check")
    hdr = os.path.join(ROOT, "raytracing_project_amd", "host", "flyscene.hpp")
    lib = os.path.join(ROOT, "raytracing_project_amd", "librtx.so")
    if not os.path.exists(lib):
        g.build()
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        g.build_host_tools()
    return exe


@pytest.fixture(scope="module")
def oa():
    from oracle import oracle_api
    oracle_api.build()
    return oracle_api


def read_dump(path):
    b = open(path, "rb").read()
    nv, nt, nm = struct.unpack_from("<3i", b, 0)
    off = 12
    verts = np.frombuffer(b, np.float32, nv * 3, off).reshape(nv, 3); off += nv * 12
    idx = np.frombuffer(b, np.uint32, nt * 3, off).reshape(nt, 3); off += nt * 12
    tm = np.frombuffer(b, np.int32, nt, off); off += nt * 4
    mats = []
    for _ in range(nm):
        f = np.frombuffer(b, np.float32, 12, off); illum = struct.unpack_from("<i", b, off + 48)[0]; off += 52
        mats.append((tuple(f[0:3]), tuple(f[3:6]), tuple(f[6:9]), f[9], f[10], f[11], illum))
    assert off == len(b)
    return verts, idx, tm, mats


def run(tool, *args):
    r = subprocess.run([tool, *map(str, args)], capture_output=True, text=True, timeout=120)
    return r.returncode, r.stderr


def same_as_oracle(tool, oa, obj, tmp_path):
    rc, err = run(tool, "obj", obj, tmp_path / "d.bin")
    assert rc == 0, err
    v, idx, tm, mats = read_dump(tmp_path / "d.bin")
    ov, oidx, otm, omats, _ = oa.obj_load(str(obj))
    assert v.tobytes() == ov.tobytes() and np.array_equal(idx, oidx) and np.array_equal(tm, otm)
    assert len(mats) == len(omats)
    for m, o in zip(mats, omats):
        assert np.array_equal(np.float32(m[0]), np.float32(tuple(o.ka))) and np.array_equal(np.float32(m[1]), np.float32(tuple(o.kd)))
        assert np.array_equal(np.float32(m[2]), np.float32(tuple(o.ks)))
        assert (np.float32(m[3]), np.float32(m[4]), np.float32(m[5]), m[6]) == (np.float32(o.ns), np.float32(o.ni), np.float32(o.d), o.illum)
    return v, idx, tm


def test_obj_loader_matches_oracle_on_a_synthetic_file(tool, oa, tmp_path):
    (tmp_path / "m.mtl").write_text("# c\nnewmtl red\nKa 0.1 0.2 0.3\nKd 1 0 0\nKs 0.5 0.5 0.5\nNs 20\nNi 1.5\nd 0.25\nillum 2\n"
                                    "\nnewmtl blue\nKd 0 0 1\n")
    (tmp_path / "m.obj").write_text("# comment\nmtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nv 1 1 1\nv 2 2 2\nvn 0 0 1\nvt 0 0\n"
                                    "usemtl blue\nf 1 2 3\nusemtl red\nf 2/1/1 3/2/2 4/3/3\nf 1//1 2//2 5//3 6//4\n"
                                    "usemtl blue\nf 4 5 6\nf 3 4 5\n")
    v, idx, tm = same_as_oracle(tool, oa, tmp_path / "m.obj", tmp_path)
    assert idx.tolist() == [[0, 1, 2], [1, 2, 3], [0, 1, 4], [3, 4, 5], [2, 3, 4]] and tm.tolist() == [1, 0, 0, 1, 1]


def test_obj_without_materials_and_missing_file(tool, oa, tmp_path):
    (tmp_path / "p.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3\nf 2 4 3\n")
    v, idx, tm = same_as_oracle(tool, oa, tmp_path / "p.obj", tmp_path)
    assert tm.tolist() == [-1, -1]                      # no usemtl: material_id stays -1
    rc, err = run(tool, "obj", tmp_path / "nope.obj", tmp_path / "d.bin")
    assert rc == 1 and "Cannot open" in err             # the reference exit(1)s here (objimporter.hpp:105)


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="the reference's bundled models are only present in the build container")
@pytest.mark.parametrize("name", ["planeobj", "cube", "twoObjects", "test", "plane", "spheres", "toy", "dodgeColorTest"])
def test_obj_loader_matches_oracle_and_goldens_on_bundled_scenes(tool, oa, tmp_path, name):
    v, idx, tm = same_as_oracle(tool, oa, os.path.join(REF_MODELS, name + ".obj"), tmp_path)
    fx = np.load(os.path.join(ROOT, "tests", "golden", "scene_%s.npz" % name))
    assert v.tobytes() == fx["verts"].tobytes() and np.array_equal(idx, fx["idx"]) and np.array_equal(tm, fx["tri_mat"])


from plyio import np_read_ply  # noqa: E402


def check_ply(tool, path, tmp_path):
    rc, err = run(tool, "ply", path, tmp_path / "p.bin")
    assert rc == 0, err
    v, idx, tm, mats = read_dump(tmp_path / "p.bin")
    ev, eidx = np_read_ply(path)
    assert v.tobytes() == ev.tobytes() and np.array_equal(idx, eidx)
    assert (tm == -1).all() and mats == []
    return v, idx


def test_ply_reader_ascii_and_binary(tool, tmp_path):
    rng = np.random.default_rng(5)
    verts = rng.standard_normal((7, 3)).astype(np.float32)
    faces = [[0, 1, 2], [2, 3, 4, 5], [6, 5, 4, 3, 2]]
    hdr = "ply\nformat %s 1.0\ncomment test\nelement vertex 7\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n" \
          "element face 3\nproperty list uchar int vertex_indices\nend_header\n"
    a = tmp_path / "a.ply"
    a.write_text(hdr % "ascii" + "".join("%r %r %r 7\n" % (float(x), float(y), float(z)) for x, y, z in verts) +
                 "".join("%d %s\n" % (len(f), " ".join(map(str, f))) for f in faces))
    b = tmp_path / "b.ply"
    with open(b, "wb") as f:
        f.write((hdr % "binary_little_endian").encode())
        for x, y, z in verts:
            f.write(struct.pack("<3fB", x, y, z, 7))
        for fc in faces:
            f.write(struct.pack("<B%di" % len(fc), len(fc), *fc))
    for path in (a, b):
        v, idx = check_ply(tool, path, tmp_path)
        assert len(v) == 7 and idx.tolist() == [[0, 1, 2], [2, 3, 4], [2, 4, 5], [6, 5, 4], [6, 4, 3], [6, 3, 2]]
    assert np.array_equal(np_read_ply(a)[0], verts)
    (tmp_path / "bad.ply").write_bytes((hdr % "binary_little_endian").encode() + b"\0" * 10)
    rc, err = run(tool, "ply", tmp_path / "bad.ply", tmp_path / "p.bin")
    assert rc == 1 and "truncated" in err


@pytest.mark.skipif(not os.path.isdir(REF_MODELS), reason="the reference's bundled models are only present in the build container")
def test_ply_reader_on_the_bundled_bunny(tool, tmp_path):
    v, idx = check_ply(tool, os.path.join(REF_MODELS, "bunny.ply"), tmp_path)
    assert len(v) == 35947 and len(idx) == 69451          # BASELINE config 2
    assert idx.max() < len(v)


def test_obj_vertex_fast_path_reads_what_the_stream_reads(tool, tmp_path):
    """loadObjFile reads plain decimal numbers with strtof instead of a std::istringstream per line (2.5x faster import);
    every other spelling goes through the stream as in Tucano's importer.  Both readings, bit for bit, on numbers in every
    spelling printf has, and the spellings that must NOT take the fast path."""
    rng = np.random.default_rng(11)
    vals = np.concatenate([rng.normal(size=3000) * 10.0 ** rng.integers(-12, 12, 3000), [0.0, -0.0, 1e-45, 1.17549435e-38, 3.4028234e38, 1e-50, 123456789.123456789]])
    fmts = ["%.9g", "%e", "%f", "%.3f", "%+.7g", "%.17g", "%.1f", "%g", "%.12e"]
    lines = []
    for k in range(0, len(vals) - 2, 3):
        f = fmts[(k // 3) % len(fmts)]
        sep = ["  ", "\t", " "][(k // 3) % 3]
        lines.append(sep.join(f % v for v in vals[k:k + 3]) + ("  " if k % 2 else ""))
    lines += [".5 5. +.25", "1 2", "7", "", "   ", "1e5 1E-5 1e+5", "-0 +0 0"]
    odd = ["1e39 2 3", "-1e39 2 3", "nan 1 2", "inf 1 2", "0x10 1 2", "1.0f 2 3", "1.5-2.5 3", "1e 2 3", "abc", "1,5 2 3", "1 2 3e", ". 1 2", "- 1 2"]
    with open(tmp_path / "v.txt", "w") as f:
        f.write("\n".join(lines + odd) + "\n")
    rc, err = run(tool, "vline", tmp_path / "v.txt", tmp_path / "v.out")
    assert rc == 0, err
    fast, slow = [int(x) for x in open(tmp_path / "v.out").read().split()[1::2]]
    assert slow >= len(odd) and fast >= len(lines) - 8, (fast, slow)


def test_ppm_writer_is_byte_identical_to_the_oracle_writer(tool, oa, tmp_path):
    from oracle.oracle_api import fp
    rng = np.random.default_rng(3)
    for n in (2, 17, 64):
        rgb = (rng.random((n, n, 3)) * 1.3 - 0.1).astype(np.float32)      # includes < 0 and > 1
        rgb[0, 1 % n] = (1.0, 0.5, 0.999)
        with open(tmp_path / "i.bin", "wb") as f:
            f.write(struct.pack("<2i", n, n)); f.write(rgb.tobytes())
        rc, err = run(tool, "ppm", tmp_path / "i.bin", tmp_path / "o.ppm")
        assert rc == 0, err
        assert oa.lib().ora_ppm_write(str(tmp_path / "e.ppm").encode(), fp(rgb), n, n, 0) == 0
        assert open(tmp_path / "o.ppm", "rb").read() == open(tmp_path / "e.ppm", "rb").read()
    # a frame large enough for the writer's bands of rows and threads, with channels far outside [0, 1] (they print as the
    # integers the reference's cast makes of them: negative numbers, 255)
    w, h = 700, 1300
    rgb = (rng.random((h, w, 3)) * 1.3 - 0.1).astype(np.float32)
    rgb[5, 7] = (-3.0e4, 7.5e5, -0.0)
    rgb[1299, 699] = (1.0, 0.0, 255.0)
    with open(tmp_path / "i.bin", "wb") as f:
        f.write(struct.pack("<2i", w, h)); f.write(rgb.tobytes())
    rc, err = run(tool, "ppm", tmp_path / "i.bin", tmp_path / "o.ppm")
    assert rc == 0, err
    assert oa.lib().ora_ppm_write(str(tmp_path / "e.ppm").encode(), fp(rgb), w, h, 0) == 0
    assert open(tmp_path / "o.ppm", "rb").read() == open(tmp_path / "e.ppm", "rb").read()
    # non-square frames (no defined behaviour in the reference): the facade's [x][y] buffer of the frame's
    # own size gives the ordinary row-major file, and the reference-signature writer refuses instead of reading outside
    for w, h in ((5, 3), (3, 8), (64, 17)):
        rgb = rng.random((h, w, 3)).astype(np.float32)
        with open(tmp_path / "i.bin", "wb") as f:
            f.write(struct.pack("<2i", w, h)); f.write(rgb.tobytes())
        rc, err = run(tool, "ppm", tmp_path / "i.bin", tmp_path / "o.ppm")
        assert rc == 0, err
        assert oa.lib().ora_ppm_write(str(tmp_path / "e.ppm").encode(), fp(rgb), w, h, 0) == 0
        assert open(tmp_path / "o.ppm", "rb").read() == open(tmp_path / "e.ppm", "rb").read()


def test_facade_mtl_loader_and_ppm_writer_match_the_reference_executed_fixture(tool, tmp_path):
    """tests/golden/io_ref.json = what the reference's OWN Tucano::MaterialImporter::loadMTL (mtlIO.hpp:45-125),
    Tucano::Material::Mtl() (mtl.hpp:21-39) and Tucano::ImageImporter::writePPMImage (ppmIO.hpp:130-151) produced
    (oracle/_ref, tests/golden/make_golden.py io).  The facade's loader and writer against them: values bit for bit,
    file bytes equal."""
    import json
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "io_ref.json")))
    for key, rec in ref["mtl"].items():
        with open(tmp_path / "m.mtl", "w", newline="") as fh:
            fh.write(rec["text"])
        # one face per material, named the way the file names it (CRLF files carry the \r in both places)
        with open(tmp_path / "m.obj", "w", newline="") as fh:
            fh.write("mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\n" + "".join("usemtl %s\nf 1 2 3\n" % n for n in rec["names"] if n))
        rc, err = run(tool, "obj", tmp_path / "m.obj", tmp_path / "d.bin")
        assert rc == 0, err
        _, _, tm, mats = read_dump(tmp_path / "d.bin")
        want = np.array(rec["values_bits"], np.uint32).reshape(-1, 13).view(np.float32)
        assert len(mats) == len(want), key
        for m, w in zip(mats, want):
            got = np.float32([*m[0], *m[1], *m[2], m[3], m[4], m[5], float(m[6])])
            assert np.array_equal(got.view(np.uint32), w.view(np.uint32)), key
        named = [n for n in rec["names"] if n]
        # (several materials of one name: the importer keeps searching and the last match wins, objimporter.hpp:145-151)
        assert tm.tolist() == [max(i for i, x in enumerate(rec["names"]) if x == n) for n in named], key
    for rec in ref["ppm"]:
        n = rec["n"]
        with open(tmp_path / "i.bin", "wb") as f:
            f.write(struct.pack("<2i", n, n)); f.write(np.array(rec["rgb_bits"], np.uint32).tobytes())
        rc, err = run(tool, "ppm", tmp_path / "i.bin", tmp_path / "o.ppm")
        assert rc == 0, err
        assert open(tmp_path / "o.ppm", newline="").read() == rec["text"], n


def test_facade_ray_value_type_matches_the_reference_goldens(tool, tmp_path):
    """Ray ctor, reflectRay, refractRay, resendRay of the facade's Ray (plain host arithmetic, like the reference's
    header-only src/Ray.h:7-58) against tests/golden/primitives.npz -- outputs of the reference's own Ray.h built with
    the vendored Eigen (oracle/_ref).  Bit for bit, NaN directions (total internal reflection) included."""
    z = np.load(os.path.join(ROOT, "tests", "golden", "primitives.npz"))
    n = len(z["o"])
    rec = np.concatenate([z["o"], z["d"], z["nq"], z["p"], z["view"], z["n12"]], axis=1).astype(np.float32)
    assert rec.shape == (n, 17)
    with open(tmp_path / "rays.bin", "wb") as f:
        f.write(struct.pack("<i", n)); f.write(np.ascontiguousarray(rec).tobytes())
    rc, err = run(tool, "ray", tmp_path / "rays.bin", tmp_path / "out.bin")
    assert rc == 0, err
    out = np.fromfile(tmp_path / "out.bin", np.float32).reshape(n, 42)
    for name, got in (("ray", out[:, 0:6]), ("reflect", out[:, 6:12]), ("refract", out[:, 12:18]), ("resend", out[:, 18:42])):
        want = z[name]
        assert np.array_equal(np.isnan(got), np.isnan(want)), name
        assert np.array_equal(got.view(np.uint32)[~np.isnan(want)], want.view(np.uint32)[~np.isnan(want)]), name
    assert np.isnan(z["refract"]).any()


def test_obj_polygon_records_follow_tucano_and_bad_ply_counts_are_refused(tool, oa, tmp_path):
    """A face record with other than three vertices: Tucano pushes every item onto the index stream and cuts triples
    afterwards, so later triangles of the group shift -- reproduced as is (face ids are the parity contract), checked
    against the oracle's loader.  A PLY list count no polygon has is refused instead of allocated."""
    with open(tmp_path / "quad.obj", "w") as f:
        f.write("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nf 1 2 3 4\nf 1 2 5\nf 1 2\n")
    v, idx, tm = same_as_oracle(tool, oa, tmp_path / "quad.obj", tmp_path)
    assert idx.tolist() == [[0, 1, 2], [3, 0, 1], [4, 0, 1]]
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\nproperty list int int vertex_indices\nend_header\n"
    body = struct.pack("<9f", 0, 0, 0, 1, 0, 0, 0, 1, 0)
    for count in (-1, 1 << 30):
        with open(tmp_path / "bad.ply", "wb") as f:
            f.write(hdr.encode()); f.write(body); f.write(struct.pack("<i", count)); f.write(struct.pack("<3i", 0, 1, 2))
        rc, err = run(tool, "ply", tmp_path / "bad.ply", tmp_path / "bad.bin")
        assert rc == 1 and "truncated PLY body" in err


def test_facade_vectors_convert_from_and_to_eigen(tmp_path):
    """With Eigen on the include path (the reference's vendored copy, where /root/reference is present) the facade's
    Vector3f takes and gives Eigen::Vector3f implicitly: the reference's call sites (Ray(origin, dir), Box fields,
    getDirection() into an Eigen variable) compile unchanged and the values agree bit for bit with Eigen's own."""
    inc = "/root/reference/dependencies/eigen/include"
    if not os.path.isdir(os.path.join(inc, "Eigen")):
        pytest.skip("no Eigen headers here (the reference's vendored copy is absent)")
    lib = os.path.join(ROOT, "raytracing_project_amd", "librtx.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    exe = str(tmp_path / "eigen_interop")
    subprocess.run(["g++", "-std=c++14", "-O2", "-ffp-contract=off", "-Wall", "-pthread", "-I" + inc, os.path.join(ROOT, "tests", "cpp", "eigen_interop.cpp"),
                    "-o", exe, "-L" + os.path.join(ROOT, "raytracing_project_amd"), "-lrtx",
                    "-Wl,-rpath," + os.path.join(ROOT, "raytracing_project_amd")], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "eigen interop ok" in r.stdout, r.stdout + r.stderr

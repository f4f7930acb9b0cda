"""Minimal numpy PLY reader: test infrastructure (checks the C++ facade's reader, feeds make_golden.py)."""
import numpy as np


def np_read_ply(path):
    """Minimal independent PLY reader (ascii / binary_little_endian, triangles or fans)."""
    b = open(path, "rb").read()
    end = b.index(b"end_header\n") + len(b"end_header\n")
    lines = b[:end].decode().split("\n")
    binary = any(l.startswith("format binary_little_endian") for l in lines)
    elems = []
    for l in lines:
        w = l.split()
        if w[:1] == ["element"]:
            elems.append([w[1], int(w[2]), []])
        elif w[:1] == ["property"]:
            elems[-1][2].append(w[1:])
    T = {"char": "i1", "uchar": "u1", "short": "<i2", "ushort": "<u2", "int": "<i4", "uint": "<u4", "float": "<f4", "double": "<f8"}
    verts, faces = [], []
    if binary:
        off = end
        for name, n, props in elems:
            if all(p[0] != "list" for p in props):
                dt = np.dtype([(p[1], T[p[0]]) for p in props])
                a = np.frombuffer(b, dt, n, off); off += n * dt.itemsize
                if name == "vertex":
                    verts = np.stack([a["x"], a["y"], a["z"]], 1).astype(np.float32)
            else:
                for _ in range(n):
                    for p in props:
                        c = int(np.frombuffer(b, T[p[1]], 1, off)[0]); off += np.dtype(T[p[1]]).itemsize
                        ids = np.frombuffer(b, T[p[2]], c, off).astype(np.int64); off += c * np.dtype(T[p[2]]).itemsize
                        if name == "face":
                            faces += [[ids[0], ids[i], ids[i + 1]] for i in range(1, c - 1)]
    else:
        tok = b[end:].split()
        k = 0
        for name, n, props in elems:
            for _ in range(n):
                rec = {}
                for p in props:
                    if p[0] != "list":
                        rec[p[1]] = float(tok[k]); k += 1
                    else:
                        c = int(tok[k]); k += 1
                        ids = [int(x) for x in tok[k:k + c]]; k += c
                        if name == "face":
                            faces += [[ids[0], ids[i], ids[i + 1]] for i in range(1, c - 1)]
                if name == "vertex":
                    verts.append([np.float32(rec["x"]), np.float32(rec["y"]), np.float32(rec["z"])])
    return np.asarray(verts, np.float32).reshape(-1, 3), np.asarray(faces, np.uint32).reshape(-1, 3)

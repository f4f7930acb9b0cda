"""MI355X-native ray-scene intersection engine for the Raytracing-Project renderer.

The product is `librtx.so` (hand-written HIP kernels + a C ABI, include/rtx.h); this
package is the thin ctypes plumbing tests and bench.py use to reach it, plus the
synthetic scene generators.  There is no CPU fallback: importing `api` raises if the
HIP library has not been built (see __graft_entry__.build()).
"""
from . import api  # noqa: F401

"""The facade's threading contract (the reference's queries "are called concurrently from 16 threads",
src/flyscene.cpp:417-430) under ThreadSanitizer, on the CPU: tests/cpp/facade_threads.cpp runs the reference's own
16-thread updating_pixels pattern and a storm of per-ray / batched queries against raytracing_project_amd/host/flyscene.hpp
linked with tests/cpp/rtx_stub.cpp, a stand-in for the C ABI in which overlapping calls on one rtx_scene are a data race.
The same threads straight onto the C ABI (no facade) are the negative control: TSan must report those."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("tsan") / "facade_threads")
    r = subprocess.run(["g++", "-std=c++14", "-O1", "-g", "-fsanitize=thread", "-Wall", os.path.join(ROOT, "tests", "cpp", "facade_threads.cpp"),
                        os.path.join(ROOT, "tests", "cpp", "rtx_stub.cpp"), "-o", out, "-lpthread"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return out


def _run(exe, *args):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66")
    # (ASLR with many mapping bits upsets this gcc's TSan runtime on new kernels: setarch -R where it is needed and available)
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=300, env=env)
    if "unexpected memory mapping" in r.stderr:
        r = subprocess.run(["setarch", os.uname().machine, "-R", exe, *args], capture_output=True, text=True, timeout=300, env=env)
    return r


def test_sixteen_threads_through_the_facade_are_race_free(exe, tmp_path):
    (tmp_path / "m.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3\nf 2 4 3\n")
    r = _run(exe, "facade", str(tmp_path / "m.obj"))
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and "facade ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_the_detector_fires_without_the_facade(exe):
    r = _run(exe, "raw")
    assert "ThreadSanitizer: data race" in r.stderr and r.returncode == 66, r.stdout[-500:] + r.stderr[-1500:]

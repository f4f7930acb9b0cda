"""bench.py's one JSON line: the keys the driver reads (and the roofline / cpu_baseline objects of the task's measurement
contract), on the smallest workload so that it runs in seconds; and the self-launching N > 1 path rehearsed on one GPU
(RTX_BENCH_REHEARSE=1: gloo, both ranks on GPU 0), whose reassembled frame must equal the full-frame render."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=280, env=e, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_contract():
    d = _run(["--workload", "planeobj_512_primary_shadow", "--steps", "24", "--warmup", "8", "--cpu-sample", "64"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 24 and d["warmup"] == 8 and d["higher_is_better"] is True
    assert d["unit"] == "Mrays/s" and d["value"] > 0 and d["vs_baseline"] is None and d["dtype"] == "f32"
    assert d["config"]["workload"] == "planeobj_512_primary_shadow" and "model" not in d["config"]
    assert abs(d["ms_per_step"] * d["value"] * 1e3 - sum(d["config"]["rays_per_frame"].values())) < 1e-3 * sum(d["config"]["rays_per_frame"].values())
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["survey_8d_frac"] > 0 and r["traffic_source"] is None or "committed" in r["traffic_source"]
    assert r["cpu_bvh_full_frame"]["rays"]["primary"] == 512 * 512 and r["cpu_bvh_full_frame"]["value"] > 0
    # median of repeated timed loops, every repeat listed; the single-frame latency figure; the standing check ran and found nothing
    rp = d["repeats"]
    assert rp["n"] == 5 and len(rp["value"]) == 5 and rp["value"] == sorted(rp["value"]) and rp["value"][2] == pytest.approx(d["value"])
    assert d["single_frame"]["ms_per_frame"] > 0 and d["single_frame"]["mrays_per_s"] > 0
    assert d["verify"]["checked"] > 0 and d["verify"]["mismatches"] == 0
    assert "camera path" in d["config"]["camera"]
    sp = d["scaling_proxy"]
    assert set(sp["render_speedup_bound"]) == {"1", "2", "4", "8"} and sp["gather_bytes_into_rank0_per_frame"]["8"]["rgb8"] * 4 == sp["gather_bytes_into_rank0_per_frame"]["8"]["f32"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1


def test_bench_starts_its_own_ranks_and_reassembles_the_frame():
    d = _run(["--gpus", "2", "--workload", "planeobj_512_primary_shadow", "--steps", "12", "--warmup", "4"], {"RTX_BENCH_REHEARSE": "1"})
    assert d["n_gpus"] == 2 and d["config"]["rehearsal"] is True
    assert d["config"]["rehearsal_equals_full_frame"] is True
    # what travelled: the P3 bytes of the frame, 3 per pixel of the other rank's tiles
    assert d["config"]["gather"]["format"] == "rgb8" and d["config"]["gather"]["bytes_into_rank0_per_frame"] == 128 * 32 * 32 * 3
    d = _run(["--gpus", "2", "--workload", "planeobj_512_primary_shadow", "--steps", "12", "--warmup", "4", "--out-format", "f32", "--repeats", "2"],
             {"RTX_BENCH_REHEARSE": "1"})
    assert d["config"]["rehearsal_equals_full_frame"] is True and d["config"]["gather"]["bytes_per_pixel"] == 12

#!/bin/bash
# scripts/quick_bench.sh <tag> : bench the three single-GPU workloads, print one line each
TAG=$1; shift
mkdir -p gpurun_out
for wl in soup1M_4k_primary_shadow bunny70k_1080p_primary_shadow hall262k_1080p_reflect3; do
  timeout -k 10 300 python3 -u bench.py --steps 32 --warmup 8 --no-scaling-proxy --no-cpu-baseline --no-roofline --no-extras --repeats 3 --workload $wl "$@" > gpurun_out/qb_${TAG}_$wl.log 2>&1 || { echo "$wl failed"; tail -5 gpurun_out/qb_${TAG}_$wl.log; exit 1; }
  python3 - <<PY
import json
l=[x for x in open("gpurun_out/qb_${TAG}_$wl.log") if x.startswith("{")][-1]
r=json.loads(l)
k=r["config"]["kernel_span_ms_per_frame_rank0"]
print("%-32s Mrays/s=%8.1f ms/step=%7.3f cam=%.3f queue=%.3f all=%.3f rays=%s" % ("$wl", r["value"], r["ms_per_step"], k["trace_camera"], k["trace_queue"], k["all_kernels"], r["config"]["rays_per_frame"]))
PY
done

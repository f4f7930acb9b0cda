#!/bin/bash
# scripts/sweep_libs.sh <lib...> : A/B of library builds (RTX_LIB) on the three single-GPU workloads.  Per build
# and workload one line: Mrays/s with the default frames per call and calls in flight, and the traversal kernels' own durations measured
# one frame at a time.  Extra environment (RTX_SCHED=...) is passed through.
mkdir -p gpurun_out
for wl in ${WORKLOADS:-soup1M_4k_primary_shadow hall262k_1080p_reflect3 bunny70k_1080p_primary_shadow}; do
for lib in "$@"; do
  tag=$(basename $lib .so)
  RTX_LIB=$PWD/$lib timeout -k 10 200 python3 -u bench.py --steps ${STEPS:-32} --warmup 2 --no-cpu-baseline --no-roofline --no-scaling-proxy --workload $wl > gpurun_out/sw_${tag}.log 2>&1 || { echo "$lib failed"; tail -5 gpurun_out/sw_${tag}.log; exit 1; }
  RTX_LIB=$PWD/$lib timeout -k 10 200 python3 -u bench.py --steps ${STEPS:-32} --warmup 2 --inflight 1 --no-cpu-baseline --no-roofline --no-scaling-proxy --workload $wl > gpurun_out/sw1_${tag}.log 2>&1 || { echo "$lib failed"; tail -5 gpurun_out/sw1_${tag}.log; exit 1; }
  python3 - <<PY
import json
r=json.loads([x for x in open("gpurun_out/sw_${tag}.log") if x.startswith("{")][-1])
r1=json.loads([x for x in open("gpurun_out/sw1_${tag}.log") if x.startswith("{")][-1])
k=r1["config"]["kernel_span_ms_per_frame_rank0"]
print("%-22s %-24s Mrays/s=%8.1f | one at a time: %8.1f cam=%.3f queue=%.3f all=%.3f" % ("$wl"[:22], "${tag}", r["value"], r1["value"], k["trace_camera"], k["trace_queue"], k["all_kernels"]))
PY
done
done

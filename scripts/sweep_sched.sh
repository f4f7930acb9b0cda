#!/bin/bash
# scripts/sweep_sched.sh : every scheduler mode forced on the three single-GPU workloads
mkdir -p gpurun_out
for wl in soup1M_4k_primary_shadow bunny70k_1080p_primary_shadow hall262k_1080p_reflect3; do
 for sc in refill split groups; do
  RTX_SCHED=$sc timeout -k 10 200 python3 -u bench.py --steps 32 --warmup 8 --no-scaling-proxy --no-cpu-baseline --no-roofline --workload $wl > gpurun_out/ss_${sc}_$wl.log 2>&1 || { echo "$wl $sc failed"; tail -5 gpurun_out/ss_${sc}_$wl.log; exit 1; }
  python3 - <<PY
import json
l=[x for x in open("gpurun_out/ss_${sc}_$wl.log") if x.startswith("{")][-1]
r=json.loads(l)
k=r["config"]["kernel_span_ms_per_frame_rank0"]
print("%-32s %-7s Mrays/s=%8.1f cam=%.3f queue=%.3f all=%.3f" % ("$wl", "$sc", r["value"], k["trace_camera"], k["trace_queue"], k["all_kernels"]))
PY
 done
done

#!/bin/bash
# scripts/sweep_libs.sh <workload> <lib...> : one bench line per library variant (forced refill scheduler)
WL=$1; shift
mkdir -p gpurun_out
for lib in "$@"; do
  tag=$(basename $lib .so)
  RTX_SCHED=refill RTX_LIB=$PWD/$lib timeout -k 10 200 python3 -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --workload $WL > gpurun_out/sw_${tag}.log 2>&1 || { echo "$lib failed"; tail -5 gpurun_out/sw_${tag}.log; exit 1; }
  python3 - <<PY
import json
l=[x for x in open("gpurun_out/sw_${tag}.log") if x.startswith("{")][-1]
r=json.loads(l)
k=r["config"]["kernel_ms_per_frame_rank0"]
print("%-22s %-28s Mrays/s=%8.1f cam=%.3f queue=%.3f all=%.3f" % ("$WL"[:22], "${tag}", r["value"], k["trace_camera"], k["trace_queue"], k["all_kernels"]))
PY
done

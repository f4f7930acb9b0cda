#!/bin/bash
# A/B of traversal variants on the GPU box: scripts/ab_bench.sh <tag> [bench args]
TAG=$1; shift
mkdir -p gpurun_out
for mode in 0 1; do
  RTX_PERSISTENT=$mode timeout -k 10 300 python3 -u bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline "$@" > gpurun_out/ab_${TAG}_p$mode.log 2>&1 || { echo "mode $mode failed"; tail -5 gpurun_out/ab_${TAG}_p$mode.log; exit 1; }
  python3 - <<PY
import json
l=[x for x in open("gpurun_out/ab_${TAG}_p$mode.log") if x.startswith("{")][-1]
r=json.loads(l)
print("persistent=$mode", "Mrays/s=%.1f"%r["value"], "ms/step=%.3f"%r["ms_per_step"], r["config"]["kernel_ms_per_frame_rank0"], r["config"]["rays_per_frame"])
PY
done

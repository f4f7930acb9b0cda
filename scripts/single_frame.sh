#!/bin/bash
# scripts/single_frame.sh <tag> [bench args]: the bench's single-frame figure (one frame per call, one call at a time) and batched value
TAG=$1; shift
for wl in soup1M_4k_primary_shadow bunny70k_1080p_primary_shadow; do
  timeout -k 10 300 python3 -u bench.py --steps 32 --warmup 8 --no-scaling-proxy --no-cpu-baseline --no-roofline --repeats 3 --workload $wl "$@" > gpurun_out/sf_${TAG}_$wl.log 2>&1 || { echo "$wl failed"; tail -5 gpurun_out/sf_${TAG}_$wl.log; exit 1; }
  python3 - <<PY
import json
r=json.loads([x for x in open("gpurun_out/sf_${TAG}_$wl.log") if x.startswith("{")][-1])
print("%-32s %-8s batched %8.1f Mrays/s  single frame %.3f ms (min %.3f) = %8.1f Mrays/s" % ("$wl", "$TAG", r["value"], r["single_frame"]["ms_per_frame"], r["single_frame"]["ms_per_frame_min"], r["single_frame"]["mrays_per_s"]))
PY
done

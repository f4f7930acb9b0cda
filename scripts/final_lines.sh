#!/bin/bash
# scripts/final_lines.sh : the bench lines committed under profiles/ (all five configs, then the soup three times with the driver's command and the default)
mkdir -p gpurun_out/final
for wl in planeobj_512_primary_shadow bunny70k_1080p_primary_shadow hall262k_1080p_reflect3 soup1M_4k_primary_shadow soup5M_4k_16l_16spp; do
  timeout -k 10 600 python3 -u bench.py --workload $wl > gpurun_out/final/$wl.log 2>&1 || { echo "$wl failed"; tail -5 gpurun_out/final/$wl.log; exit 1; }
  grep '^{' gpurun_out/final/$wl.log | tail -n 1 > gpurun_out/final/r03_bench_$wl.json
  echo "$wl done"
done
: > gpurun_out/final/consecutive.jsonl
for i in 1 2 3; do
  timeout -k 10 400 python3 -u bench.py --steps 20 --warmup 5 > gpurun_out/final/c.log 2>&1 || { tail -5 gpurun_out/final/c.log; exit 1; }
  grep '^{' gpurun_out/final/c.log | tail -n 1 >> gpurun_out/final/consecutive.jsonl
  timeout -k 10 400 python3 -u bench.py > gpurun_out/final/c.log 2>&1 || { tail -5 gpurun_out/final/c.log; exit 1; }
  grep '^{' gpurun_out/final/c.log | tail -n 1 >> gpurun_out/final/consecutive.jsonl
  echo "pair $i done"
done

#!/bin/bash
# scripts/proxy_sweep.sh <lib> : 1/N-share frame times (scaling proxy) for frames in flight x units per wave
for d in ${DEPTHS:-3 4 6 8}; do for u in ${UPWS:-1 8 16}; do
RTX_UPW=$u RTX_LIB=$PWD/$1 timeout -k 10 200 python3 -u bench.py --steps 30 --warmup 2 --inflight $d --no-cpu-baseline --no-roofline --workload ${WL:-soup1M_4k_primary_shadow} > gpurun_out/ps.log 2>&1 || { tail -3 gpurun_out/ps.log; exit 1; }
python3 - <<PY
import json
r=json.loads([x for x in open("gpurun_out/ps.log") if x.startswith("{")][-1])
print("inflight $d upw $u: full %.3f ms" % r["ms_per_step"], {k: round(v,3) for k,v in r["scaling_proxy"]["ms_per_frame_part0_of_N"].items()})
PY
done; done

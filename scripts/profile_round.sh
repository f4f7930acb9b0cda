#!/bin/bash
# Runs on the GPU box (via gpurun): every rocprofv3 pass behind profiles/rNN_<workload>.txt and rNN_pmc.json.
# Usage: scripts/profile_round.sh <tag> <workload> [extra bench args]; writes gpurun_out/<tag>_<workload>.txt
# One frame at a time (--inflight 1), so that a kernel's counters and durations are its own.
set -u
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
ARGS="--workload $WL --inflight 1 --no-scaling-proxy $*"
bash scripts/profile_gpu.sh ${TAG}_$WL $ARGS > gpurun_out/${TAG}_$WL.plog 2>&1 || { tail -5 gpurun_out/${TAG}_$WL.plog; exit 1; }
bash scripts/profile_mem.sh ${TAG}_$WL $ARGS >> gpurun_out/${TAG}_$WL.plog 2>&1 || { tail -5 gpurun_out/${TAG}_$WL.plog; exit 1; }
OUT=gpurun_out/${TAG}_$WL.txt
python3 scripts/prof_summary.py gpurun_out/prof_${TAG}_$WL > $OUT
python3 scripts/prof_summary.py gpurun_out/mem_${TAG}_$WL >> $OUT
python3 scripts/prof_derive.py $OUT >> $OUT.derived && cat $OUT.derived >> $OUT && rm $OUT.derived
echo "command: python3 bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-roofline --no-extras --repeats 1 $ARGS" >> $OUT
# (--steps 16 --inflight 1 with the default batch: one call of 16 frames per timed repeat; RTX_PROF_FRAMES says otherwise)
python3 scripts/prof_derive.py $OUT --pmc-json $WL --frames-per-launch ${RTX_PROF_FRAMES:-16} > gpurun_out/${TAG}_${WL}_pmc.json
tail -8 $OUT
# the rocprofv3 databases are tens of MB per pass; what is wanted is in the summaries (gpurun copies back at most 64 MiB)
rm -rf gpurun_out/prof_${TAG}_$WL gpurun_out/mem_${TAG}_$WL

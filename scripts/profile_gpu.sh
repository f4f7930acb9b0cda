#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace + PMC passes of bench.py.
# Usage: scripts/profile_gpu.sh <tag> [bench args...]; results under gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
BENCH="python3 $ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-roofline --no-extras --repeats 1 $*"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace pass failed"; tail -5 $OUT/trace.log; exit 1; }
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 300 rocprofv3 --pmc $pass -d $OUT/pmc_$name -o pmc -- $BENCH > $OUT/pmc_$name.log 2>&1 || { echo "pmc pass $pass failed"; tail -5 $OUT/pmc_$name.log; }
done
ls -R $OUT | head -50

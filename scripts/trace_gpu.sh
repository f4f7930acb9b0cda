#!/bin/bash
# kernel trace only: scripts/trace_gpu.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
python3 scripts/prof_summary.py $OUT | grep -E "rtx::|kernel " | head -20

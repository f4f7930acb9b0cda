#!/bin/bash
# scripts/pmc_quick.sh <tag> [bench args]: one SQ counter pass + kernel trace of the soup bench (one call at a time), derived per-kernel lines
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
BENCH="python3 $ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-roofline --no-extras --repeats 1 --inflight 1 --no-scaling-proxy $*"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $BENCH > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE -d $OUT/pmc_sq -o pmc -- $BENCH > $OUT/pmc_sq.log 2>&1 || { echo "pmc failed"; tail -5 $OUT/pmc_sq.log; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_tcc -o pmc -- $BENCH > $OUT/pmc_tcc.log 2>&1 || { echo "pmc tcc failed"; tail -5 $OUT/pmc_tcc.log; exit 1; }
python3 scripts/prof_summary.py $OUT > gpurun_out/pq_$TAG.txt
python3 scripts/prof_derive.py gpurun_out/pq_$TAG.txt >> gpurun_out/pq_$TAG.txt
rm -rf $OUT
head -9 gpurun_out/pq_$TAG.txt | cut -c1-150
grep -A12 "== derived" gpurun_out/pq_$TAG.txt | grep "k_trace\|k_shade\|k_resolve"

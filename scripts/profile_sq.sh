#!/bin/bash
# SQ-side PMC passes of bench.py (issue mix, stalls): scripts/profile_sq.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
BENCH="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extras --repeats 1 $*"
i=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_BRANCH" \
            "SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_FLAT"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass -d $OUT/pmc_$i -o pmc -- $BENCH > $OUT/pmc_$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc_$i.log; exit 1; }
done
python3 $ROOT/scripts/prof_summary.py $OUT 2>/dev/null | grep "k_trace\|k_shade" 

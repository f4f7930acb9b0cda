#!/bin/bash
# L1-side (TA/TCP/TD) PMC passes of bench.py: scripts/profile_mem.sh <tag> [bench args]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/mem_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
BENCH="python3 $ROOT/bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-roofline --no-extras --repeats 1 $*"
i=0
for pass in "TA_TA_BUSY_sum TA_BUSY_avr TD_TD_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
            "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE" \
            "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TD_TC_STALL_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $pass -d $OUT/pmc_$i -o pmc -- $BENCH > $OUT/pmc_$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc_$i.log; exit 1; }
done
python3 $ROOT/scripts/prof_summary.py $OUT 2>/dev/null | grep "k_trace" 

#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of scripts/build_study.py (device builds, rebuilds, refits of one workload).
# Usage: scripts/profile_build.sh <workload>; writes gpurun_out/build_<workload>.txt
set -u
WL=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_build_$WL/trace -o trace -- python3 $ROOT/scripts/build_study.py $WL > $ROOT/gpurun_out/prof_build_$WL.log 2>&1 || { tail -5 $ROOT/gpurun_out/prof_build_$WL.log; exit 1; }
cd $ROOT
OUT=gpurun_out/build_$WL.txt
grep '^{' gpurun_out/prof_build_$WL.log > $OUT
python3 scripts/prof_summary.py gpurun_out/prof_build_$WL >> $OUT
echo "command: rocprofv3 --kernel-trace --stats -- python3 scripts/build_study.py $WL   (1 first build + 6 rebuilds on the device; 8 refits of the device-built tree, 9 of the host-built one)" >> $OUT
rm -rf gpurun_out/prof_build_$WL
tail -25 $OUT

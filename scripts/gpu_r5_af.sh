#!/bin/bash
# Kernel timelines of the 4M-row mixture call at tolerance 0.01 on either schedule (what the iteration behind the
# hand-over point costs, launch by launch).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5af}
python scripts/config_b.py --samples 200000 --verbosity 0 > /dev/null 2>&1
for yy in 0.1 0; do
rm -rf $OUT/prof_$TAG
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_$TAG -o p -- python scripts/config_b.py --samples 4000000 --data gaussian --yinyang $yy --verbosity 1 > $OUT/prof_$TAG.log 2>&1
grep -E "iteration|Lloyd|wall" $OUT/prof_$TAG.log | tr '\r' '\n' | grep -v "^step" | tail -12 > $OUT/timeline_yy${yy}_$TAG.log
python scripts/kernel_timeline.py $OUT/prof_$TAG/p_results.db 120 >> $OUT/timeline_yy${yy}_$TAG.log
rm -rf $OUT/prof_$TAG
done

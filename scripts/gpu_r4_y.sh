#!/bin/bash
# Round 4: pair certificates.   bash scripts/gpu_r4_y.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4y2}
timeout 600 python -m pytest tests/test_gpu_carry.py -m gpu -q 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head
timeout 200 python scripts/stress_carry.py 120 7 > $OUT/stress_carry_$TAG.log 2>&1; tail -1 $OUT/stress_carry_$TAG.log; grep -E "^FAIL|^ERR" $OUT/stress_carry_$TAG.log | head; grep -c "paired [1-9]" $OUT/stress_carry_$TAG.log
echo "== kernel trace of the 4M-row mixture call, default schedule"
rm -rf $OUT/prof_mix_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_mix_$TAG -o p -- python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0 > $OUT/prof_mix_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_mix_$TAG/p_results.db $OUT/kernel_stats_mixture_$TAG.csv | head -3 | cut -c1-60
rm -rf $OUT/prof_mix_$TAG

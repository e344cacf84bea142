#!/bin/bash
# Round 4: carry_skip_kernel with one cursor atomic per block.   bash scripts/gpu_r4_x.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4x}
timeout 600 python -m pytest tests/test_gpu_carry.py -m gpu -q 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head
run() { echo "## $1" | tee -a $OUT/configs_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|kmeans_cuda\(" | tee -a $OUT/configs_$TAG.log; }
: > $OUT/configs_$TAG.log
run "4M-row mixture tol 1e-4: default (verbosity 2 for the spared count)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 2
run "4M-row mixture tol 1e-4: default, silent" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "angular 4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 0
echo "== carry trace: rows stage 2 takes per pass"
KMCUDA_AMD_CARRY_TRACE=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0 2>&1 | grep -E "stage 2 took|\] pass" | sed -n '20,32p' | cut -c1-260
echo "== kernel trace of the 4M-row mixture call, default schedule"
rm -rf $OUT/prof_mix_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_mix_$TAG -o p -- python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0 > $OUT/prof_mix_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_mix_$TAG/p_results.db $OUT/kernel_stats_mixture_$TAG.csv | head -8 | cut -c1-60,200-300
rm -rf $OUT/prof_mix_$TAG

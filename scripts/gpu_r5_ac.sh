#!/bin/bash
# Config B and the 4M-row mixture at tolerance 0.01: the default schedule against yinyang_t = 0, interleaved on one box,
# and what the carry policy saw in config B (KMCUDA_AMD_CARRY_TRACE: list lengths per pass; that run synchronises).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5ac}
run() { echo "## $1" | tee -a $OUT/short_calls_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall" | tee -a $OUT/short_calls_$TAG.log; }
python scripts/config_b.py --samples 200000 --verbosity 0 > /dev/null 2>&1
for i in 1 2 3; do
run "config B yinyang_t=0.1" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B yinyang_t=0" timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
run "4M mixture tol 0.01 yinyang_t=0.1" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0
run "4M mixture tol 0.01 yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0
done
KMCUDA_AMD_CARRY_TRACE=1 timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 1 2>&1 | grep -E "\[carry\] pass|paused|Lloyd goes|carrying" | cut -c1-200 > $OUT/carry_trace_config_b_$TAG.log
for i in 1 2; do
run "4M mixture tol 1e-4 yinyang_t=0.1" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "2M x 384 uniform yinyang_t=0.1" timeout 300 python scripts/config_b.py --samples 2000000 --features 384 --verbosity 0
done

#!/bin/bash
# Round 4: pair certificates under both metrics.   bash scripts/gpu_r4_y.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4y4}
timeout 300 python -m pytest tests/test_gpu_carry.py -m gpu -q 2>&1 | grep -E "passed|failed|^E  |^FAILED" | head
run() { echo "## $1" | tee -a $OUT/configs_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|carried pairs|kmeans_cuda\(" | tee -a $OUT/configs_$TAG.log; }
: > $OUT/configs_$TAG.log
run "angular 4M-row mixture tol 1e-4: default (verbosity 2)" timeout 100 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 2
run "angular 4M-row mixture tol 1e-4: default, silent" timeout 100 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "angular 4M-row mixture tol 1e-4: KMCUDA_AMD_CARRY_PAIRS=0" env KMCUDA_AMD_CARRY_PAIRS=0 timeout 100 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 0

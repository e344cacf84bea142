#!/bin/bash
# Round 4: carried-bounds tests, where the group clustering's time goes (KMCUDA_AMD_TIMING), config B / mixture walls.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4c}
echo "== carry tests"
timeout 900 python -m pytest tests/test_gpu_carry.py -m gpu -q > $OUT/pytest_carry_$TAG.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_carry_$TAG.log
echo "== mixture with timing"
KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0 2>&1 | grep -E "timing|kmeans_cuda wall" | tee $OUT/timing_$TAG.log
KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0 2>&1 | grep -E "timing|kmeans_cuda wall" | tee -a $OUT/timing_$TAG.log
echo "== config B: default (carry) / KMCUDA_AMD_CARRY=0 / yinyang_t=0"
( timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 2 | grep -E "kmeans_cuda wall|carried bounds"
  KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0 | grep -o "kmeans_cuda wall.*"
  timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0 | grep -o "kmeans_cuda wall.*" ) 2>&1 | tee $OUT/configB_$TAG.log
echo "== mixture tol 1e-4: default / yinyang_t=0"
( timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 2 | grep -E "kmeans_cuda wall|carried bounds"
  timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.0001 --verbosity 0 | grep -o "kmeans_cuda wall.*" ) 2>&1 | tee $OUT/mixture_$TAG.log

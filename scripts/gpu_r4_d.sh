#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4d}
echo "== carry tests + yinyang suite (host loop restructured)"
timeout 900 python -m pytest tests/test_gpu_carry.py tests/test_gpu_yinyang.py tests/test_gpu_sharded.py -m gpu -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?"; tail -8 $OUT/pytest_$TAG.log
echo "== mixture with timing (slab allocator)"
for i in 1 2; do KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0 2>&1 | grep -E "timing|kmeans_cuda wall"; done | tee $OUT/timing_$TAG.log
echo "-- yinyang_t=0"
KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0 2>&1 | grep -E "timing|kmeans_cuda wall" | tee -a $OUT/timing_$TAG.log
echo "== config B default / yinyang_t=0"
( timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0 | grep -o "kmeans_cuda wall.*"
  timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0 | grep -o "kmeans_cuda wall.*" ) 2>&1 | tee $OUT/configB_$TAG.log
echo "== overflow diagnosis"
timeout 1200 python scripts/overflow_diag.py 2>&1 | grep -v amdgpu.ids | tee $OUT/overflow_diag_$TAG.log

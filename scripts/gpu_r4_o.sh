#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
for pre in 1 0; do
echo "== KMCUDA_AMD_PRELOAD=$pre: mixture yinyang_t=0 (one call per process), three processes"
for i in 1 2 3; do KMCUDA_AMD_PRELOAD=$pre timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0 2>&1 | grep -E "kmeans_cuda wall"; done
echo "== KMCUDA_AMD_PRELOAD=$pre: config B yinyang_t=0"
KMCUDA_AMD_PRELOAD=$pre timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0 2>&1 | grep -E "kmeans_cuda wall"
done | tee $OUT/preload_ab.log
KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0 2>&1 | grep -E "timing\] [a-zA-Z]|iteration 1 judged|kmeans_cuda wall" | tee -a $OUT/preload_ab.log
timeout 600 python -m pytest tests/test_gpu_kmeans.py -m gpu -q -x 2>&1 | tail -2

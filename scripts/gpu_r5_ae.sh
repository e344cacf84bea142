#!/bin/bash
# Where a short call's time goes on either schedule (KMCUDA_AMD_TIMING laps; each lap waits for the GPU): the 4M-row
# mixture at tolerance 0.01, yinyang_t = 0.1 against 0.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5ae}
python scripts/config_b.py --samples 200000 --verbosity 0 > /dev/null 2>&1
for i in 1 2 3; do for yy in 0.1 0; do
echo "## yinyang_t=$yy" | tee -a $OUT/laps_$TAG.log
KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang $yy --verbosity 0 2>&1 | grep -E "timing|kmeans_cuda wall" | tee -a $OUT/laps_$TAG.log
done; done

#!/bin/bash
# Round 4: the angular carried bound (per-centroid bias change), the listed pass striding, reports judged once.
#   bash scripts/gpu_r4_t.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4t}
timeout 900 python -m pytest tests/test_gpu_carry.py tests/test_gpu_kmeans.py tests/test_gpu_fp16.py tests/test_gpu_yinyang.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -4
run() { echo "## $1" | tee -a $OUT/configs_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|knn_cuda|calculated|kmeans_cuda\(" | tee -a $OUT/configs_$TAG.log; }
: > $OUT/configs_$TAG.log
for rep in 1 2; do
run "angular 4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "angular 4M-row mixture tol 1e-4: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
done
run "angular 4M-row mixture tol 1e-4: default (verbosity 2)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 2
run "angular fp16 4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --dtype f16 --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "angular fp16 4M-row mixture tol 1e-4: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --dtype f16 --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "config B 8Mx256 K=1024 tol 0.01: default" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B: yinyang_t=0" timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
run "config C shape: fp16 angular, 8 virtual 1M-row shards: default" env KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0
run "config C shape: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0

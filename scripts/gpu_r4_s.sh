#!/bin/bash
# Round 4: after the listed pass learnt to stride and the carried bounds the angular metric -- regression of the
# suites around them and the whole calls they touch.   bash scripts/gpu_r4_s.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4s}
timeout 900 python -m pytest tests/test_gpu_carry.py tests/test_gpu_yinyang.py tests/test_gpu_kmeans.py tests/test_gpu_fp16.py tests/test_gpu_sharded.py tests/test_gpu_golden.py -m gpu -q 2>&1 | tail -4
run() { echo "## $1" | tee -a $OUT/configs_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|knn_cuda|calculated|kmeans_cuda\(" | tee -a $OUT/configs_$TAG.log; }
: > $OUT/configs_$TAG.log
run "4M-row mixture tol 1e-4: default (verbosity 2 for the spared count)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 2
run "4M-row mixture tol 1e-4: default, silent" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "config B 8Mx256 K=1024 tol 0.01: yinyang_t=0.1 default" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "angular 4M-row mixture tol 1e-4: default (verbosity 2)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 2
run "angular 4M-row mixture tol 1e-4: default, silent" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "angular 4M-row mixture tol 1e-4: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "config C shape: fp16 angular, 8 virtual 1M-row shards: default" env KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0
run "config C shape: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0

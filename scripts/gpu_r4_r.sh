#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
echo "## config D whole (8M queries) as eight virtual shards on one GPU" | tee $OUT/configD_8v_r4r.log
KMCUDA_AMD_KNN_STATS=1 KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 600 python scripts/config_d.py --samples 8000000 --check 200 2>&1 | grep -E "knn_cuda|brute|calculated" | tee -a $OUT/configD_8v_r4r.log
echo "## config D whole, one shard" | tee -a $OUT/configD_8v_r4r.log
timeout 600 python scripts/config_d.py --samples 8000000 2>&1 | grep -E "knn_cuda|calculated" | tee -a $OUT/configD_8v_r4r.log

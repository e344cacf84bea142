#!/bin/bash
# k-NN, the whole config on ONE GPU (8M queries): 8-wave blocks (the built library) against 4-wave blocks.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5r}
for lib in "" scratch/libKMCUDA_knn4w.so; do
  echo "## KMCUDA_AMD_LIB=$lib" | tee -a $OUT/knn_whole_$TAG.log
  KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} KMCUDA_AMD_KNN_STATS=1 timeout 300 python scripts/config_d.py --samples 8000000 2>&1 | grep -E "knn_cuda|k-NN filter" | cut -c1-230 | tee -a $OUT/knn_whole_$TAG.log
  KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 2>&1 | grep -E "knn_cuda" | tee -a $OUT/knn_whole_$TAG.log
  KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python scripts/config_d.py --samples 2000000 2>&1 | grep -E "knn_cuda" | tee -a $OUT/knn_whole_$TAG.log
done

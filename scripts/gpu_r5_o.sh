#!/bin/bash
# k-NN filter: operand sets none of whose queries visits the cluster are not multiplied -- A/B against the build before.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5o}
timeout 600 python -m pytest tests/test_gpu_knn.py -m gpu -q -x > $OUT/pytest_knn_$TAG.log 2>&1; echo "pytest knn rc=$?"; tail -2 $OUT/pytest_knn_$TAG.log
for lib in "" scratch/libKMCUDA_knn_before_set_skip.so "" scratch/libKMCUDA_knn_before_set_skip.so; do
  echo "## KMCUDA_AMD_LIB=$lib" | tee -a $OUT/knn_set_skip_$TAG.log
  KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} KMCUDA_AMD_KNN_STATS=1 timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 --check 32 2>&1 | grep -E "knn_cuda|brute|k-NN filter" | cut -c1-260 | tee -a $OUT/knn_set_skip_$TAG.log
done

#!/bin/bash
# k-NN filter, 8-wave blocks: the SIMDs' second waves sift their scores half a phase late (KNN16_SKEW, the built library)
# against the build without (scratch/libKMCUDA_knn_noskew.so): parity tests, then config D's share, interleaved.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5ai}
timeout 600 python -m pytest tests/test_gpu_knn.py -m gpu -q -x > $OUT/pytest_knn_$TAG.log 2>&1; echo "pytest knn rc=$?"; tail -2 $OUT/pytest_knn_$TAG.log
for lib in "" knn_noskew "" knn_noskew; do
  echo "## KMCUDA_AMD_LIB=$lib" | tee -a $OUT/knn_skew_$TAG.log
  KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/scratch/libKMCUDA_$lib.so} KMCUDA_AMD_KNN_STATS=1 timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 --check 32 2>&1 | grep -E "knn_cuda|brute|k-NN filter" | cut -c1-260 | tee -a $OUT/knn_skew_$TAG.log
done

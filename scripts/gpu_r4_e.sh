#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4e}
echo "== k-means++ filter bisect"
timeout 1500 python scripts/kmpp_filter_bisect.py 2>&1 | grep -v amdgpu.ids | tee $OUT/kmpp_bisect_$TAG.log
echo "== k-NN: parity with the query order, config D share A/B"
timeout 900 python -m pytest tests/test_gpu_knn.py -m gpu -q -x > $OUT/pytest_knn_$TAG.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_knn_$TAG.log
CMD="python scripts/config_d.py --samples 8000000 --shard 0/8 --check 100"
( KMCUDA_AMD_KNN_STATS=1 timeout 300 $CMD 2>&1 | grep -E "knn_cuda|k-NN filter|brute"
  KMCUDA_AMD_KNN_ORDER=0 KMCUDA_AMD_KNN_STATS=1 timeout 300 $CMD 2>&1 | grep -E "knn_cuda|k-NN filter|brute" | sed 's/^/ORDER=0: /' ) | tee $OUT/configD_$TAG.log

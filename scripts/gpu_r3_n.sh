#!/bin/bash
# k-NN with 256 < D <= 512 on the f16 filter: parity tests, then a timing against the exact search
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3n}
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_knn.py > $OUT/pytest_${TAG}.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_${TAG}.log
for mode in filter exact; do
  if [ $mode = exact ]; then export KMCUDA_AMD_KNN_EXACT=1; else unset KMCUDA_AMD_KNN_EXACT; fi
  timeout 900 python scripts/config_d.py --samples 400000 --features 512 --clusters 256 --check 200 > $OUT/cfgd_${TAG}_$mode.log 2>&1
  echo "$mode: $(grep -i "knn_cuda\|wall\|fraction" $OUT/cfgd_${TAG}_$mode.log | tr '\n' ' ')"
done

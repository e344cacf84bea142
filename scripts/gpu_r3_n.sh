#!/bin/bash
# k-NN with 256 < D <= 1024 on the f16 filter: parity tests, then timings (and, for 512, against the exact search)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3n}
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_knn.py > $OUT/pytest_${TAG}.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_${TAG}.log
for d in 512 768 1024; do
  timeout 900 python scripts/config_d.py --samples 400000 --features $d --clusters 256 --check 200 > $OUT/cfgd_${TAG}_$d.log 2>&1
  echo "D=$d: $(grep -iE "knn_cuda|brute|calculated" $OUT/cfgd_${TAG}_$d.log | tr '\n' ' ')"
done
timeout 900 python scripts/config_d.py --samples 1000000 --features 768 --clusters 1024 --check 200 > $OUT/cfgd_${TAG}_1M768.log 2>&1
echo "1M x 768, K=1024: $(grep -iE "knn_cuda|brute|calculated" $OUT/cfgd_${TAG}_1M768.log | tr '\n' ' ')"
if [ "${EXACT:-0}" = 1 ]; then
  KMCUDA_AMD_KNN_EXACT=1 timeout 900 python scripts/config_d.py --samples 400000 --features 512 --clusters 256 > $OUT/cfgd_${TAG}_exact.log 2>&1
  echo "exact 512: $(grep -iE "knn_cuda" $OUT/cfgd_${TAG}_exact.log)"
fi

#!/bin/bash
# Round 3, session F: k-NN heaps in LDS (tests + config D share timing, both heap homes), scale tests C / D.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3f}
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_knn.py > $OUT/pytest_${TAG}_knn.log 2>&1
echo "knn pytest rc=$?"; tail -4 $OUT/pytest_${TAG}_knn.log
echo "== config D share (LDS heap / global heap)"
timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 2>&1 | grep -E "knn_cuda|calculated"
KMCUDA_AMD_KNN_LDS_HEAP=0 timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 2>&1 | grep -E "knn_cuda|calculated"
timeout 1500 python -m pytest -q -s -m gpu tests/test_gpu_scale.py -k "config_c or config_d" > $OUT/pytest_${TAG}_scale.log 2>&1
echo "scale pytest rc=$?"; grep -E "bounds after|filter pass|answered|brute force|passed|failed|Error|error|assert" $OUT/pytest_${TAG}_scale.log | tail -12

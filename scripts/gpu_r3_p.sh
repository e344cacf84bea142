#!/bin/bash
# filtered k-means++ steps: parity tests, then timing of kmeans_cuda(init="k-means++") with / without the filter
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3p}
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_kmeans.py -k "kmeanspp or pins or plus" > $OUT/pytest_${TAG}.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest_${TAG}.log
for n in 8000000 1000000; do
for f in 1 0 1; do
  echo "N=$n filter=$f: $(KMCUDA_AMD_DEBUG=2 KMCUDA_AMD_KMPP_FILTER=$f timeout 600 python scripts/config_b.py --samples $n --init k-means++ --yinyang 0 --verbosity 0 2>&1 | grep -E "wall|exact chains|host chooser" | tr '\n' ' ')"
done
done

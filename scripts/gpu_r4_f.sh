#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4f}
echo "== k-means++ filter at the sizes that failed"
timeout 900 python scripts/kmpp_filter_bisect.py tiled 2>&1 | grep -v amdgpu.ids | tail -3 | tee $OUT/kmpp_bisect_$TAG.log
echo "== overflow pin"
timeout 900 python -m pytest tests/test_gpu_scale.py -k "overflow" -m gpu -q -x -s 2>&1 | grep -E "EXACT_UPDATE|passed|failed|assert" | tee $OUT/overflow_$TAG.log
echo "== seeding tests (random init restated, k-means++), pins"
timeout 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_golden.py -m gpu -q -x > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_$TAG.log
echo "== timing: mixture default / yinyang_t=0, config B default / yinyang_t=0"
for y in 0.1 0; do KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang $y --verbosity 0 2>&1 | grep -E "timing|kmeans_cuda wall"; done | tee $OUT/timing_$TAG.log
for y in 0.1 0; do KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --yinyang $y --verbosity 0 2>&1 | grep -E "timing|kmeans_cuda wall"; done | tee -a $OUT/timing_$TAG.log
echo "== config A: 100000 x 256 host arrays, K = 1024, tolerance 0.002 (three calls)"
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tee $OUT/configA_$TAG.log
import time, numpy
from kmcuda_amd import kmeans_cuda
numpy.random.seed(0)
x = numpy.random.rand(100000, 256).astype(numpy.float32)
for i in range(4):
    t = time.perf_counter()
    c, a = kmeans_cuda(x, 1024, init="random", seed=3, tolerance=0.002, yinyang_t=0, device=1, verbosity=0)
    print("kmeans_cuda(100000 x 256, K = 1024): %.4f s" % (time.perf_counter() - t), flush=True)
PY

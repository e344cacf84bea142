#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4g}
echo "== timing: mixture default / yinyang_t=0"
for y in 0.1 0; do KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang $y --verbosity 0 2>&1 | grep -E "timing|kmeans_cuda wall"; done | tee $OUT/timing_$TAG.log
echo "== config A with timing"
KMCUDA_AMD_TIMING=1 timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tail -30 | tee $OUT/configA_$TAG.log
import time, numpy
from kmcuda_amd import kmeans_cuda
numpy.random.seed(0)
x = numpy.random.rand(100000, 256).astype(numpy.float32)
for i in range(3):
    t = time.perf_counter()
    c, a = kmeans_cuda(x, 1024, init="random", seed=3, tolerance=0.002, yinyang_t=0, device=1, verbosity=0)
    print("kmeans_cuda(100000 x 256, K = 1024): %.4f s" % (time.perf_counter() - t), flush=True)
PY
echo "== overflow pin + yinyang pins"
timeout 900 python -m pytest tests/test_gpu_scale.py -k "overflow" tests/test_gpu_yinyang.py -m gpu -q -x 2>&1 | tail -3

#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python scripts/transpose_bench.py 2>&1 | grep transpose | tee $OUT/transpose_r4n.log
echo "== config A with timing laps (calls 2 and 3)"
KMCUDA_AMD_TIMING=1 timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | grep -E "timing\] [a-zA-Z]|kmeans_cuda\(|iteration 1 judged|iteration 2 judged|iteration 15 judged" | tail -26 | tee $OUT/configA_timing_r4n.log
import time, numpy
from kmcuda_amd import kmeans_cuda
numpy.random.seed(0)
x = numpy.random.rand(100000, 256).astype(numpy.float32)
for i in range(3):
    t = time.perf_counter()
    c, a = kmeans_cuda(x, 1024, init="random", seed=3, tolerance=0.002, yinyang_t=0, device=1, verbosity=0)
    print("kmeans_cuda(100000 x 256, K = 1024): %.4f s" % (time.perf_counter() - t), flush=True)
PY

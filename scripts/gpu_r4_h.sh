#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4h}
echo "== tests: carry + rng, kmeans, yinyang"
timeout 900 python -m pytest tests/test_gpu_carry.py tests/test_gpu_kmeans.py tests/test_gpu_yinyang.py -m gpu -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest_$TAG.log
echo "== timing: mixture default / yinyang_t=0; config B default / yinyang_t=0 (no timing laps)"
for y in 0.1 0; do KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang $y --verbosity 0 2>&1 | grep -E "timing\] [a-zA-Z]|kmeans_cuda wall"; done | tee $OUT/timing_$TAG.log
for y in 0.1 0; do timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang $y --verbosity 0 2>&1 | grep -E "kmeans_cuda wall"; done | tee -a $OUT/timing_$TAG.log
for y in 0.1 0; do timeout 300 python scripts/config_b.py --yinyang $y --verbosity 0 2>&1 | grep -E "kmeans_cuda wall"; done | tee -a $OUT/timing_$TAG.log
echo "== config A"
timeout 300 python - <<'PY' 2>&1 | grep -v amdgpu | tail -8 | tee $OUT/configA_$TAG.log
import time, numpy
from kmcuda_amd import kmeans_cuda
numpy.random.seed(0)
x = numpy.random.rand(100000, 256).astype(numpy.float32)
for i in range(5):
    t = time.perf_counter()
    c, a = kmeans_cuda(x, 1024, init="random", seed=3, tolerance=0.002, yinyang_t=0, device=1, verbosity=0)
    print("kmeans_cuda(100000 x 256, K = 1024): %.4f s" % (time.perf_counter() - t), flush=True)
PY
echo "== kernel timeline of the mixture's Lloyd call (first iteration)"
rm -rf $OUT/prof_$TAG
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_$TAG -o p -- python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0 > /dev/null 2>&1
python scripts/rocpd_timeline.py $OUT/prof_$TAG/p_results.db 0 400 kmx | head -120 | tee $OUT/timeline_mixture_$TAG.log | head -90
rm -rf $OUT/prof_$TAG

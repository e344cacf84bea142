#!/bin/bash
# GPU session B: CU timeline of the stage-1 kernel + ablations on a fixed steady state.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r2b}
timeout 600 python -m pytest tests/test_gpu_kmeans.py -x -q -k "update or fused" > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_$TAG.log
echo "== CU timeline"
KMCUDA_AMD_LIB=scratch/libs/libtrace.so timeout 300 python scripts/coarse_trace.py 8000000 $OUT/trace_$TAG.npy > $OUT/trace_$TAG.log 2>&1; echo "rc=$?"; grep -v amdgpu.ids $OUT/trace_$TAG.log | head -60
echo "== ablations on a fixed state"
timeout 300 python scripts/coarse_ab.py --save /tmp/state.npz
for v in base abl1 abl2 abl3 abl7 abl8 book1; do
  KMCUDA_AMD_LIB=scratch/libs/lib$v.so timeout 120 python scripts/coarse_ab.py --load /tmp/state.npz 2>&1 | grep coarse
done

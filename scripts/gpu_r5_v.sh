#!/bin/bash
# 257..512 features: both filters (plain passes streamed, carried passes register-resident): the affected tests, then whole kmeans_cuda() calls (default
# schedule, yinyang_t = 0.1) at 384 / 512 features with and without KMCUDA_AMD_WIDE_MIN_D=513 (the register-resident
# filter + carried bounds, the default until now).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5v}
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_carry.py -m gpu -x -q 2>&1 | tail -5 | tee $OUT/tests_$TAG.log
run() { echo "## $1" | tee -a $OUT/mid_widths_calls_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall" | tee -a $OUT/mid_widths_calls_$TAG.log; }
python scripts/config_b.py --samples 200000 --features 384 --verbosity 0 > /dev/null 2>&1   # (the box's first process)
for wide in 257 513 257; do
export KMCUDA_AMD_WIDE_MIN_D=$wide
run "2M x 384 @ 1024 uniform tol 0.01, MIN_D=$wide" timeout 300 python scripts/config_b.py --samples 2000000 --features 384 --verbosity 0
run "2M x 384 @ 1024 mixture tol 0.01, MIN_D=$wide" timeout 300 python scripts/config_b.py --samples 2000000 --features 384 --data gaussian --verbosity 0
run "2M x 384 @ 1024 mixture tol 1e-4, MIN_D=$wide" timeout 300 python scripts/config_b.py --samples 2000000 --features 384 --data gaussian --tolerance 0.0001 --verbosity 0
run "2M x 512 @ 1024 mixture tol 1e-4, MIN_D=$wide" timeout 300 python scripts/config_b.py --samples 2000000 --features 512 --data gaussian --tolerance 0.0001 --verbosity 0
done

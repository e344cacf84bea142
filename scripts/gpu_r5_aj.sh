#!/bin/bash
# Carried bounds on the streamed filter (rows wider than 512 features): parity tests, then whole calls on 2M x 1024 @ 1024
# mixtures with and without the bounds, and 257..512 features with the bounds on either filter.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5aj}
timeout 1200 python -m pytest tests/test_gpu_wide.py tests/test_gpu_carry.py -m gpu -q -x > $OUT/pytest_wide_carry_$TAG.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_wide_carry_$TAG.log | cut -c1-220
run() { echo "## $1" | tee -a $OUT/wide_carry_calls_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall" | tee -a $OUT/wide_carry_calls_$TAG.log; }
python scripts/config_b.py --samples 200000 --features 1024 --verbosity 0 > /dev/null 2>&1
for i in 1 2; do
for cv in 1x 0; do
[ $cv = 0 ] && export KMCUDA_AMD_CARRY=0 || unset KMCUDA_AMD_CARRY
run "2M x 1024 @ 1024 mixture tol 1e-4, CARRY=${KMCUDA_AMD_CARRY:-default}" timeout 300 python scripts/config_b.py --samples 2000000 --features 1024 --data gaussian --tolerance 0.0001 --verbosity 0
run "2M x 1024 @ 1024 mixture tol 0.01, CARRY=${KMCUDA_AMD_CARRY:-default}" timeout 300 python scripts/config_b.py --samples 2000000 --features 1024 --data gaussian --verbosity 0
run "2M x 1024 @ 1024 uniform tol 0.01, CARRY=${KMCUDA_AMD_CARRY:-default}" timeout 300 python scripts/config_b.py --samples 2000000 --features 1024 --verbosity 0
done
unset KMCUDA_AMD_CARRY
for f in streamed; do
:
run "2M x 384 @ 1024 mixture tol 1e-4, bounds on the $f filter" timeout 300 python scripts/config_b.py --samples 2000000 --features 384 --data gaussian --tolerance 0.0001 --verbosity 0
run "2M x 512 @ 1024 mixture tol 1e-4, bounds on the $f filter" timeout 300 python scripts/config_b.py --samples 2000000 --features 512 --data gaussian --tolerance 0.0001 --verbosity 0
run "2M x 384 @ 1024 uniform tol 0.01, bounds on the $f filter" timeout 300 python scripts/config_b.py --samples 2000000 --features 384 --verbosity 0
done
unset KMCUDA_AMD_CARRY_FILTER
done

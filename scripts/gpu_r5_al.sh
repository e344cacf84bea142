#!/bin/bash
# Angular pair certificates off by default: the carry / wide / kmeans tests, the randomised whole-call stress again.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5al}
timeout 900 python -m pytest tests/test_gpu_carry.py tests/test_gpu_wide.py -m gpu -q > $OUT/pytest_carry_wide_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_carry_wide_$TAG.log | cut -c1-200
timeout 200 python scripts/stress_carry_api.py 75 58 > $OUT/stress_carry_api_$TAG.log 2>&1; echo "stress api rc=$?"; grep -c "^ok" $OUT/stress_carry_api_$TAG.log; grep -E "FAIL|ERR|trials" $OUT/stress_carry_api_$TAG.log | head

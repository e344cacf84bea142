#!/bin/bash
# Round 4, second session: carried bounds -- parity tests, then what they buy on the two benchmark data sets, whole
# kmeans_cuda() calls with device-resident rows (config B as named; the 4M-row 1024-Gaussian mixture), against
# yinyang_t = 0 and against the same schedule with plain passes (KMCUDA_AMD_CARRY=0).   bash scripts/gpu_r4_b.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4b}
echo "== carry tests"
timeout 900 python -m pytest tests/test_gpu_carry.py -m gpu -q -x > $OUT/pytest_carry_$TAG.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest_carry_$TAG.log
if [ "${SKIP_REST:-0}" = 1 ]; then exit 0; fi
echo "== pins / schedules"
timeout 900 python -m pytest tests/test_gpu_yinyang.py -m gpu -q -x -k "15_3 or schedules or 9" > $OUT/pytest_yy_$TAG.log 2>&1; echo "rc=$?"; tail -5 $OUT/pytest_yy_$TAG.log
echo "== 4M-row Gaussian mixture: default (carry) / KMCUDA_AMD_CARRY=0 / yinyang_t=0 / reference schedule"
( timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 2 | grep -E "kmeans_cuda wall|carried bounds|iteration" | tail -40
  KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0 | grep -o "kmeans_cuda wall.*"
  timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0 | grep -o "kmeans_cuda wall.*"
  KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0 | grep -o "kmeans_cuda wall.*" ) 2>&1 | tee $OUT/mixture_$TAG.log
echo "== same, tolerance 1e-4"
( timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 2 | grep -E "kmeans_cuda wall|carried bounds"
  timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.0001 --verbosity 0 | grep -o "kmeans_cuda wall.*" ) 2>&1 | tee -a $OUT/mixture_$TAG.log
echo "== config B: default (carry) / KMCUDA_AMD_CARRY=0 / yinyang_t=0"
( timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 2 | grep -E "kmeans_cuda wall|carried bounds"
  KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0 | grep -o "kmeans_cuda wall.*"
  timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0 | grep -o "kmeans_cuda wall.*" ) 2>&1 | tee $OUT/configB_$TAG.log
echo "== bench (the dense Lloyd step must not have moved)"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; python3 -c "
import json;d=json.loads(open('$OUT/bench_$TAG.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['kernel_ms'], d.get('verify',{}).get('ok'))"
echo "== overflow pin"
timeout 900 python -m pytest tests/test_gpu_scale.py -k "overflow" -m gpu -q -x -s 2>&1 | grep -E "EXACT_UPDATE|passed|failed" | tee $OUT/overflow_$TAG.log

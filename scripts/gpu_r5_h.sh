#!/bin/bash
# After the update's path choice was fixed (largest list unknown after a radix call -> radix again): the update tests,
# the short calls that used to lose at the hand-over point, the bench.   bash scripts/gpu_r5_h.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5h}
timeout 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_lloyd.py tests/test_gpu_exact_update.py -m gpu -q -x > $OUT/pytest_update_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_update_$TAG.log
echo "== whole calls" | tee $OUT/configs_$TAG.log
run() { echo "## $1" | tee -a $OUT/configs_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|timing\] (Lloyd|group)" | tee -a $OUT/configs_$TAG.log; }
for rep in 1 2; do
run "4M-row mixture tol 0.01: default (yinyang_t=0.1)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.01 --verbosity 0
run "4M-row mixture tol 0.01: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.01 --verbosity 0
run "config B: default" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B: yinyang_t=0" timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
done
run "config B: default, timing laps" env KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY

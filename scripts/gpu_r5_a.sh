#!/bin/bash
# Round 5, first evidence run: the oracle inside the carried-bounds tests (every pass of every carrying loop, 1M-row
# passes, one whole call under the strict update), the bench line with its fixed diagnostics, the 1M-row shard, and
# what the strict switches of INTEGRATION.md 6 cost on config B.   bash scripts/gpu_r5_a.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5a}
timeout 900 python -m pytest tests/test_gpu_carry.py -m gpu -q -x --durations=8 > $OUT/pytest_carry_$TAG.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_carry_$TAG.log
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; head -c 600 $OUT/bench_$TAG.json; echo
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step","rows_full_exact_scan_last_step","rows_pair_refined_last_step","reassigned_last_step")}, d["roofline"]["kernel_ms"], d["roofline"]["traffic"], d["roofline"].get("traffic_source"), d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY
echo "== 1M-row shard"
timeout 300 python bench.py --samples 1000000 --steps 40 --warmup 10 --no-cpu-baseline --verify-rows 200000 > $OUT/bench1m_$TAG.json 2> $OUT/bench1m_$TAG.err; echo "rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench1m_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY
echo "== config B under the strict switches (INTEGRATION.md 6)" | tee $OUT/configs_strict_$TAG.log
run() { echo "## $1" | tee -a $OUT/configs_strict_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds" | tee -a $OUT/configs_strict_$TAG.log; }
run "config B default (yinyang_t=0.1)" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B KMCUDA_AMD_EXACT_UPDATE=1 KMCUDA_AMD_YY=carry (strict update, default schedule)" env KMCUDA_AMD_EXACT_UPDATE=1 KMCUDA_AMD_YY=carry timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B KMCUDA_AMD_EXACT_UPDATE=1 yinyang_t=0 (strict update, Lloyd)" env KMCUDA_AMD_EXACT_UPDATE=1 timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
run "config B KMCUDA_AMD_YY=reference (reference schedule, default update)" env KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B KMCUDA_AMD_EXACT_UPDATE=1 KMCUDA_AMD_YY=reference (the reference end to end)" env KMCUDA_AMD_EXACT_UPDATE=1 KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0

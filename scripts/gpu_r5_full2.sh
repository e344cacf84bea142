#!/bin/bash
# The whole GPU suite + smoke + the bench line, then the randomised carried-vs-plain stress runs (oracle in every pass).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5full2}
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu_full_$TAG.log 2>&1; echo "pytest rc=$?"; tail -22 $OUT/pytest_gpu_full_$TAG.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["roofline"]["traffic_source"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"), d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
timeout 400 python scripts/stress_carry.py 200 57 > $OUT/stress_carry_$TAG.log 2>&1; echo "stress rc=$?"; tail -3 $OUT/stress_carry_$TAG.log
timeout 300 python scripts/stress_carry_api.py 60 58 > $OUT/stress_carry_api_$TAG.log 2>&1; echo "stress api rc=$?"; tail -3 $OUT/stress_carry_api_$TAG.log
for shape in "2000000 1024 100000" "4000000 384 100000"; do set -- $shape
timeout 300 python bench.py --samples $1 --features $2 --steps 10 --warmup 5 --no-cpu-baseline --no-api-leg --verify-rows $3 > $OUT/bench_wide_${1}x${2}_$TAG.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/bench_wide_${1}x${2}_$TAG.json').read().strip().splitlines()[-1])
print('$1 x $2', {k:d[k] for k in ('value','ms_per_step')}, d['breakdown_ms_per_step'], d.get('verify',{}).get('ok'))" | tee -a $OUT/wide_shapes_$TAG.log
done

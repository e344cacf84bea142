#!/bin/bash
# Round 3, session E: refine kernel without scratch, scale tests of configs C / D, everything touched since.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3e}
timeout 1500 python -m pytest -q -s -m gpu tests/test_gpu_scale.py -k "config_c or config_d" > $OUT/pytest_${TAG}_scale.log 2>&1
echo "scale pytest rc=$?"; grep -E "bounds after|filter pass|answered|brute force|passed|failed|Error|error|assert" $OUT/pytest_${TAG}_scale.log | tail -12
timeout 1500 python -m pytest -q -m gpu tests/test_gpu_lloyd.py tests/test_gpu_row_cache.py tests/test_gpu_kmeans.py tests/test_gpu_fp16.py tests/test_gpu_golden.py tests/test_gpu_sharded.py tests/test_gpu_scale.py -k "not config_c and not config_d" > $OUT/pytest_${TAG}.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_${TAG}.log
for rep in 1 2; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $OUT/bench_${TAG}_8M_$rep.json 2>> $OUT/bench_${TAG}.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_${TAG}_8M_$rep.json"))
print("8M $rep", d["ms_per_step"], d["breakdown_ms_per_step"], d["roofline"]["frac"], d["roofline"]["filter_stage_ms"]-d["roofline"]["kernel_ms"])
PY
done
timeout 200 python bench.py --samples 1000000 --steps 40 --warmup 10 --no-cpu-baseline --verify-rows 200000 > $OUT/bench_${TAG}_1M.json 2>>$OUT/bench_${TAG}.err
python - <<PY
import json
d=json.load(open("$OUT/bench_${TAG}_1M.json"))
print("1M", {k:d[k] for k in ("value","ms_per_step","breakdown_ms_per_step")}, d["verify"]["ok"])
PY

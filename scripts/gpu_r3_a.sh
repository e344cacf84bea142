#!/bin/bash
# Round 3, session A: the device-side stop rule / persistent shard workers / adaptive Yinyang schedule.
# Targeted tests, then the bench line, the 1M-row shard, whole kmeans_cuda() calls with 1 and 8 (virtual)
# shards, config B both ways.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3a}
timeout 900 python -m pytest -q -x -m gpu tests/test_gpu_kmeans.py tests/test_gpu_yinyang.py tests/test_gpu_sharded.py \
   tests/test_gpu_exact_update.py tests/test_gpu_fp16.py tests/test_gpu_golden.py > $OUT/pytest_${TAG}.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/pytest_${TAG}.log
echo "== bench 8M"
timeout 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_${TAG}_8M.json 2> $OUT/bench_${TAG}_8M.err; tail -c 400 $OUT/bench_${TAG}_8M.err
python - <<PY
import json
d=json.load(open("$OUT/bench_${TAG}_8M.json"))
print({k:d[k] for k in ("value","ms_per_step","breakdown_ms_per_step")}, d["roofline"]["frac"], d["verify"]["ok"])
PY
echo "== bench 1M shard"
timeout 200 python bench.py --samples 1000000 --steps 40 --warmup 10 --no-cpu-baseline --verify-rows 200000 > $OUT/bench_${TAG}_1M.json 2>>$OUT/bench_${TAG}_8M.err
python - <<PY
import json
d=json.load(open("$OUT/bench_${TAG}_1M.json"))
print({k:d[k] for k in ("value","ms_per_step","breakdown_ms_per_step")}, d["verify"]["ok"])
PY
echo "== api 8M, 1 shard"
timeout 200 python bench.py --api --steps 20 > $OUT/bench_${TAG}_api1.json 2>>$OUT/bench_${TAG}_8M.err
python -c "import json;d=json.load(open('$OUT/bench_${TAG}_api1.json'));print(d['ms_per_step'], d['calls'])"
echo "== api 8M, 8 virtual shards"
KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 200 python bench.py --api --steps 20 > $OUT/bench_${TAG}_api8v.json 2>>$OUT/bench_${TAG}_8M.err
python -c "import json;d=json.load(open('$OUT/bench_${TAG}_api8v.json'));print(d['ms_per_step'], d['calls'])"
echo "== api 8M, 8 virtual shards, no speculation"
KMCUDA_AMD_SPECULATE=0 KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 200 python bench.py --api --steps 20 > $OUT/bench_${TAG}_api8v_nospec.json 2>>$OUT/bench_${TAG}_8M.err
python -c "import json;d=json.load(open('$OUT/bench_${TAG}_api8v_nospec.json'));print(d['ms_per_step'], d['calls'])"
echo "== api 1M, 1 shard (what one of 8 GPUs runs)"
timeout 200 python bench.py --api --samples 1000000 --steps 20 > $OUT/bench_${TAG}_api1M.json 2>>$OUT/bench_${TAG}_8M.err
python -c "import json;d=json.load(open('$OUT/bench_${TAG}_api1M.json'));print(d['ms_per_step'], d['calls'])"
echo "== config B (yinyang_t=0.1 default schedule / reference schedule / Lloyd)"
timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0 2>&1 | grep -o "kmeans_cuda wall.*"
KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0 2>&1 | grep -o "kmeans_cuda wall.*"
timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0 2>&1 | grep -o "kmeans_cuda wall.*"
echo "== gaussian mixture 4M (default / reference / Lloyd)"
timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0 2>&1 | grep -o "kmeans_cuda wall.*"
KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0 2>&1 | grep -o "kmeans_cuda wall.*"
timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0 2>&1 | grep -o "kmeans_cuda wall.*"

#!/bin/bash
# Round 3, session C: lloyd_settle, apply_prepare, rank sort.  Tests that touch them, the bench lines, a kernel
# trace of the 1M-row shard.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3c}
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_lloyd.py tests/test_gpu_kmeans.py tests/test_gpu_sharded.py \
   tests/test_gpu_row_cache.py tests/test_gpu_golden.py tests/test_gpu_exact_update.py tests/test_gpu_yinyang.py -k "not many_passes" > $OUT/pytest_${TAG}.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/pytest_${TAG}.log
echo "== bench 8M"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_${TAG}_8M.json 2> $OUT/bench_${TAG}_8M.err; tail -c 300 $OUT/bench_${TAG}_8M.err
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
echo "== api 8M 1 shard / 8 virtual / 1M"
timeout 200 python bench.py --api --steps 20 > $OUT/bench_${TAG}_api1.json 2>>$OUT/bench_${TAG}_8M.err
python -c "import json;d=json.load(open('$OUT/bench_${TAG}_api1.json'));print(d['ms_per_step'], [c['loop_s'] for c in d['calls']])"
KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 200 python bench.py --api --steps 20 > $OUT/bench_${TAG}_api8v.json 2>>$OUT/bench_${TAG}_8M.err
python -c "import json;d=json.load(open('$OUT/bench_${TAG}_api8v.json'));print(d['ms_per_step'], [c['loop_s'] for c in d['calls']])"
timeout 200 python bench.py --api --samples 1000000 --steps 20 --tolerance 0.0001 > $OUT/bench_${TAG}_api1M.json 2>>$OUT/bench_${TAG}_8M.err
python -c "import json;d=json.load(open('$OUT/bench_${TAG}_api1M.json'));print(d['ms_per_step'], [(c['iterations'],c['loop_s']) for c in d['calls']])"
echo "== rocprof 1M shard"
rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_1M -o p -- python bench.py --samples 1000000 --steps 20 --warmup 10 --no-cpu-baseline --no-verify > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/prof_${TAG}_1M/p_results.db $OUT/kernel_stats_${TAG}_1M.csv > /dev/null 2>&1
rm -rf $OUT/prof_${TAG}_1M
echo "== rocprof 8M"
rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_8M -o p -- python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-verify > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/prof_${TAG}_8M/p_results.db $OUT/kernel_stats_${TAG}_8M.csv > /dev/null 2>&1
rm -rf $OUT/prof_${TAG}_8M

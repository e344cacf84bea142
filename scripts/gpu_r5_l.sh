#!/bin/bash
# After the update's path rule was put right (an older report stays valid; no marker): update tests, --api twice, the
# whole calls, the wide bench with two slices in flight in its contender stage, config D whole on one GPU.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5l}
timeout 900 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_lloyd.py tests/test_gpu_wide.py tests/test_gpu_sharded.py -m gpu -q -x > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_$TAG.log
for rep in 1 2; do
timeout 600 python bench.py --api --steps 20 > $OUT/bench_api_${TAG}_$rep.json 2> $OUT/bench_api_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api_${TAG}_$rep.json'));print('api', d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])"
done
KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 600 python bench.py --api --gpus 8 --steps 20 > $OUT/bench_api8v_$TAG.json 2> $OUT/bench_api8v_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api8v_$TAG.json'));print('api 8 virtual', d['ms_per_step'], d['config']['collective_ms_per_step'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])"
echo "== whole calls" | tee $OUT/configs_$TAG.log
run() { echo "## $1" | tee -a $OUT/configs_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|knn_cuda" | tee -a $OUT/configs_$TAG.log; }
for rep in 1 2; do
run "4M-row mixture tol 0.01: default (yinyang_t=0.1)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.01 --verbosity 0
run "4M-row mixture tol 0.01: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.01 --verbosity 0
run "config B: default" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B: yinyang_t=0" timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
done
run "4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.0001 --verbosity 0
run "config C shape (fp16 angular, 8 virtual shards), k-means++" env KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --init k-means++ --metric cos --dtype f16 --yinyang 0.1 --verbosity 0
run "config D whole on one GPU (8M queries)" timeout 600 python scripts/config_d.py --samples 8000000
echo "== wide rows"
timeout 300 python bench.py --samples 2000000 --features 1024 --steps 10 --warmup 5 --no-cpu-baseline --verify-rows 100000 > $OUT/bench_wide_$TAG.json 2> $OUT/bench_wide_$TAG.err; echo "rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_wide_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY

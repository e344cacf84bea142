#!/bin/bash
# Round 4, evidence session: the whole GPU suite + smoke, the bench line, its kernel trace and PMC passes (timed
# iterations only), the 1M-row shard, whole kmeans_cuda() calls with 1 and 8 (virtual) shards, BASELINE configs A / B /
# C / D at their named sizes, the carried-bounds schedule against plain Lloyd.   bash scripts/gpu_r4_final.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4z}
if [ "${SKIP_SUITE:-0}" != 1 ]; then
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_full_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_full_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; head -c 1200 $OUT/bench_$TAG.json; echo
echo "== rocprofv3 kernel trace of the same command"
rm -rf $OUT/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_$TAG/p_results.db $OUT/kernel_stats_$TAG.csv | head -12 | cut -c1-150
rm -rf $OUT/prof_$TAG
echo "== 1M-row shard"
timeout 300 python bench.py --samples 1000000 --steps 40 --warmup 10 --no-cpu-baseline --verify-rows 200000 > $OUT/bench1m_$TAG.json 2> $OUT/bench1m_$TAG.err; echo "rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench1m_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY
echo "== --api: whole kmeans_cuda() calls, 1 shard / 8 virtual shards / one 1M-row shard"
timeout 600 python bench.py --api --steps 20 > $OUT/bench_api_$TAG.json 2> $OUT/bench_api_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api_$TAG.json'));print(d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])"
KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 600 python bench.py --api --steps 20 > $OUT/bench_api8v_$TAG.json 2>> $OUT/bench_api_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api8v_$TAG.json'));print(d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])"
timeout 600 python bench.py --api --samples 1000000 --steps 20 --tolerance 0.0001 > $OUT/bench_api1m_$TAG.json 2>> $OUT/bench_api_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api1m_$TAG.json'));print(d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4)) for c in d['calls']])"
echo "== PMC (timed iterations)"
bash scripts/gpu_pmc_all.sh $TAG 2>&1 | tail -6
echo "== whole calls (verbosity 0, device-resident rows)" | tee $OUT/configs_$TAG.log
run() { echo "## $1" | tee -a $OUT/configs_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|knn_cuda|calculated|k-NN filter|kmeans_cuda\(" | tee -a $OUT/configs_$TAG.log; }
run "config B 8Mx256 K=1024 tol 0.01: yinyang_t=0.1 default" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B: KMCUDA_AMD_YY=reference" env KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B: yinyang_t=0" timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
run "4M-row mixture tol 0.01: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0
run "4M-row mixture tol 0.01: reference" env KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0
run "4M-row mixture tol 0.01: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0
run "4M-row mixture tol 1e-4: default (verbosity 2 for the spared count)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 2
run "4M-row mixture tol 1e-4: default, silent" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: reference" env KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.0001 --verbosity 0
run "config C shape: fp16 angular, 8 virtual 1M-row shards: default" env KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0
run "config C shape: reference schedule" env KMCUDA_AMD_YY=reference KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0
run "config D share: 1M queries of rank 0 of 8 against the 8Mx256 corpus" env KMCUDA_AMD_KNN_STATS=1 timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8
echo "## transpose against HBM" | tee -a $OUT/configs_$TAG.log; timeout 300 python scripts/transpose_bench.py 2>&1 | grep transpose | tee -a $OUT/configs_$TAG.log
echo "## config A: 100000 x 256 host arrays, K = 1024, tolerance 0.002, five calls in one process" | tee -a $OUT/configs_$TAG.log
timeout 300 python - <<'PY' 2>&1 | grep "kmeans_cuda(" | tee -a $OUT/configs_$TAG.log
import time, numpy
from kmcuda_amd import kmeans_cuda
numpy.random.seed(0)
x = numpy.random.rand(100000, 256).astype(numpy.float32)
for i in range(5):
    t = time.perf_counter()
    c, a = kmeans_cuda(x, 1024, init="random", seed=3, tolerance=0.002, yinyang_t=0, device=1, verbosity=0)
    print("kmeans_cuda(100000 x 256, K = 1024): %.4f s" % (time.perf_counter() - t), flush=True)
PY

#!/bin/bash
# Round 5 evidence on the final build: bench (twice), its kernel trace, the PMC passes, the 1M-row shard, --api, config D
# over eight virtual shards with k-means++ seeding.   bash scripts/gpu_r5_k.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5k}
for rep in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_${TAG}_$rep.json 2> $OUT/bench_${TAG}_$rep.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_${TAG}_$rep.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY
done
echo "== rocprofv3 kernel trace of the same command"
rm -rf $OUT/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_$TAG/p_results.db $OUT/kernel_stats_$TAG.csv | head -9 | cut -c1-150
rm -rf $OUT/prof_$TAG
echo "== PMC passes"
bash scripts/gpu_pmc_all.sh $TAG 2>&1 | tail -8 | cut -c1-400
echo "== 1M-row shard"
timeout 300 python bench.py --samples 1000000 --steps 40 --warmup 10 --no-cpu-baseline --verify-rows 200000 > $OUT/bench1m_$TAG.json 2> $OUT/bench1m_$TAG.err; echo "rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench1m_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY
echo "== --api"
timeout 600 python bench.py --api --steps 20 > $OUT/bench_api_$TAG.json 2> $OUT/bench_api_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api_$TAG.json'));print(d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])"
echo "== config D, eight virtual shards, k-means++ seeding" | tee $OUT/configD_8v_$TAG.log
KMCUDA_AMD_VIRTUAL_SHARDS=8 KMCUDA_AMD_KNN_STATS=1 timeout 600 python scripts/config_d.py --samples 8000000 --init k-means++ --check 64 2>&1 | grep -E "knn_cuda|brute|k-NN filter" | cut -c1-250 | tee -a $OUT/configD_8v_$TAG.log

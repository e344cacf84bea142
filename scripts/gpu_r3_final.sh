#!/bin/bash
# Round 3, evidence session: the whole GPU suite + smoke, the bench line, its kernel trace and PMC passes (timed
# iterations only), the 1M-row shard, whole kmeans_cuda() calls with 1 and 8 (virtual) shards, two self-launched
# ranks over gloo, BASELINE configs B / C / D at their named sizes.   bash scripts/gpu_r3_final.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3z}
if [ "${SKIP_SUITE:-0}" != 1 ]; then
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_full_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_full_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
fi
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; head -c 1800 $OUT/bench_$TAG.json; echo
echo "== rocprofv3 kernel trace of the same command"
rm -rf $OUT/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_$TAG/p_results.db $OUT/kernel_stats_$TAG.csv | head -14 | cut -c1-160
rm -rf $OUT/prof_$TAG
echo "== 1M-row shard (+ kernel trace)"
timeout 300 python bench.py --samples 1000000 --steps 40 --warmup 10 --no-cpu-baseline --verify-rows 200000 > $OUT/bench1m_$TAG.json 2> $OUT/bench1m_$TAG.err; echo "rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench1m_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof1m_$TAG -o p -- python bench.py --samples 1000000 --steps 20 --warmup 10 --no-cpu-baseline --no-verify > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/prof1m_$TAG/p_results.db $OUT/kernel_stats_1M_$TAG.csv > /dev/null; rm -rf $OUT/prof1m_$TAG
echo "== --api: whole kmeans_cuda() calls, 1 shard / 8 virtual shards / one 1M-row shard"
timeout 600 python bench.py --api --steps 20 > $OUT/bench_api_$TAG.json 2> $OUT/bench_api_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api_$TAG.json'));print(d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])"
KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 600 python bench.py --api --steps 20 > $OUT/bench_api8v_$TAG.json 2>> $OUT/bench_api_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api8v_$TAG.json'));print(d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])"
timeout 600 python bench.py --api --samples 1000000 --steps 20 --tolerance 0.0001 > $OUT/bench_api1m_$TAG.json 2>> $OUT/bench_api_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api1m_$TAG.json'));print(d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4)) for c in d['calls']])"
echo "== --gpus 2 on one device over gloo (self-launching path)"
KMCUDA_AMD_BENCH_SINGLE_DEVICE=1 KMCUDA_AMD_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 5 --warmup 3 --samples 2000000 --no-cpu-baseline > $OUT/bench2_$TAG.json 2> $OUT/bench2_$TAG.err; echo "rc=$?"; head -c 500 $OUT/bench2_$TAG.json; echo; tail -2 $OUT/bench2_$TAG.err
echo "== PMC (timed iterations)"
bash scripts/gpu_pmc_all.sh $TAG 2>&1 | tail -8
echo "== config B: yinyang_t=0.1 default schedule / reference schedule / yinyang_t=0"
( timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0; KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0; timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0 ) 2>&1 | grep -o "kmeans_cuda wall.*" | tee $OUT/configB_$TAG.log
echo "== 4M-row Gaussian mixture: default / reference / Lloyd"
( timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0; KMCUDA_AMD_YY=reference timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --verbosity 0; timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0 ) 2>&1 | grep -o "kmeans_cuda wall.*" | tee -a $OUT/configB_$TAG.log
echo "== config C shape: fp16 angular, 8 virtual shards, default / reference schedule"
( KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0; KMCUDA_AMD_YY=reference KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0 ) 2>&1 | grep -o "kmeans_cuda wall.*" | tee $OUT/configC_$TAG.log
echo "== config D share"
timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 2>&1 | grep -E "knn_cuda|calculated" | tee $OUT/configD_$TAG.log

#!/bin/bash
# Round 4, closing evidence on the last commit (what changed since scripts/gpu_r4_final.sh's r4zz run: the listed pass
# strides, the angular carried bound, how list reports are judged, the preparation kernel's eighth statistics word):
# every GPU test file but the two long ones whose code did not change (test_gpu_scale.py, test_gpu_knn.py), smoke, the
# bench line + its kernel trace, the 1M-row shard, one kmeans_cuda() bench, the whole calls.  bash scripts/gpu_r4_final2.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4v}
timeout 900 python -m pytest tests/test_gpu_carry.py tests/test_gpu_yinyang.py tests/test_gpu_kmeans.py tests/test_gpu_fp16.py tests/test_gpu_sharded.py tests/test_gpu_golden.py tests/test_gpu_lloyd.py tests/test_gpu_row_cache.py tests/test_gpu_exact_update.py tests/test_gpu_wide.py -m gpu -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; head -c 700 $OUT/bench_$TAG.json; echo
echo "== rocprofv3 kernel trace of the same command"
rm -rf $OUT/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_$TAG/p_results.db $OUT/kernel_stats_$TAG.csv | head -8 | cut -c1-150
rm -rf $OUT/prof_$TAG
echo "== 1M-row shard"
timeout 300 python bench.py --samples 1000000 --steps 40 --warmup 10 --no-cpu-baseline --verify-rows 200000 > $OUT/bench1m_$TAG.json 2> $OUT/bench1m_$TAG.err; echo "rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench1m_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY
echo "== --api: whole kmeans_cuda() calls"
timeout 600 python bench.py --api --steps 20 > $OUT/bench_api_$TAG.json 2> $OUT/bench_api_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api_$TAG.json'));print(d['ms_per_step'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])"
echo "== whole calls (verbosity 0, device-resident rows)" | tee $OUT/configs_$TAG.log
run() { echo "## $1" | tee -a $OUT/configs_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|knn_cuda|calculated|k-NN filter|kmeans_cuda\(" | tee -a $OUT/configs_$TAG.log; }
run "config B 8Mx256 K=1024 tol 0.01: yinyang_t=0.1 default" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B: yinyang_t=0" timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
run "4M-row mixture tol 1e-4: default (verbosity 2 for the spared count)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 2
run "4M-row mixture tol 1e-4: default, silent" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: KMCUDA_AMD_CARRY=0" env KMCUDA_AMD_CARRY=0 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.0001 --verbosity 0
run "angular 4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "angular 4M-row mixture tol 1e-4: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0 --tolerance 0.0001 --verbosity 0
run "config C shape: fp16 angular, 8 virtual 1M-row shards: default" env KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0
run "config D share: 1M queries of rank 0 of 8 against the 8Mx256 corpus" env KMCUDA_AMD_KNN_STATS=1 timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8
echo "== kernel trace of the 4M-row mixture call, default schedule"
rm -rf $OUT/prof_mix_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_mix_$TAG -o p -- python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0 > $OUT/prof_mix_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_mix_$TAG/p_results.db $OUT/kernel_stats_mixture_$TAG.csv | head -14 | cut -c1-150
rm -rf $OUT/prof_mix_$TAG
echo "== whole calls at random: default schedule against KMCUDA_AMD_CARRY=0"
timeout 200 python scripts/stress_carry_api.py 70 11 > $OUT/stress_carry_api_$TAG.log 2>&1; tail -1 $OUT/stress_carry_api_$TAG.log; grep -E "^FAIL|^ERR" $OUT/stress_carry_api_$TAG.log | head

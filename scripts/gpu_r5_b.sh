#!/bin/bash
# Round 5, second run: k-means++ over row shards (seeds == one shard == oracle), the collective's own clock in both
# bench loops, seeding times at 8M rows with 1 and 8 (virtual) shards.   bash scripts/gpu_r5_b.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5b}
timeout 1200 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_sharded.py -m gpu -q -x --durations=8 > $OUT/pytest_kmpp_$TAG.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_kmpp_$TAG.log
echo "== seeding times (KMCUDA_AMD_TIMING laps), 8M x 256, K = 1024, init = k-means++, tolerance 0.5" | tee $OUT/kmpp_times_$TAG.log
run() { echo "## $1" | tee -a $OUT/kmpp_times_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|\[timing\] (seeding|set-up)|host chooser" | tee -a $OUT/kmpp_times_$TAG.log; }
run "one shard" env KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --init k-means++ --tolerance 0.5 --yinyang 0 --verbosity 0
run "8 virtual shards" env KMCUDA_AMD_TIMING=1 KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --init k-means++ --tolerance 0.5 --yinyang 0 --verbosity 0
run "8 virtual shards, host chooser (round 4's path)" env KMCUDA_AMD_TIMING=1 KMCUDA_AMD_VIRTUAL_SHARDS=8 KMCUDA_AMD_KMPP_HOST=1 timeout 400 python scripts/config_b.py --init k-means++ --tolerance 0.5 --yinyang 0 --verbosity 0 --clusters 128
run "one shard, K = 128 (for the line above)" env KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --init k-means++ --tolerance 0.5 --yinyang 0 --verbosity 0 --clusters 128
run "8 virtual shards, K = 128" env KMCUDA_AMD_TIMING=1 KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --init k-means++ --tolerance 0.5 --yinyang 0 --verbosity 0 --clusters 128
run "config C shape (fp16 angular, 8 virtual shards), k-means++" env KMCUDA_AMD_TIMING=1 KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --init k-means++ --metric cos --dtype f16 --yinyang 0.1 --verbosity 0
echo "== --api, 8 virtual shards: the library's clock around its all-reduce stand-in"
KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 600 python bench.py --api --gpus 8 --steps 20 > $OUT/bench_api8v_$TAG.json 2> $OUT/bench_api8v_$TAG.err; python -c "import json;d=json.load(open('$OUT/bench_api8v_$TAG.json'));print(d['ms_per_step'], d['config']['collective_ms_per_step'], d['config']['ranks_seen_by_communicator'], [(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3)) for c in d['calls']])"

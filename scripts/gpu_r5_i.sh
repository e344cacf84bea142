#!/bin/bash
# Round 5: the angular k-means++ chooser on the device (listed outliers + the host's roundings replayed).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5i}
timeout 1200 python -m pytest tests/test_gpu_kmeans.py -m gpu -q -x --durations=6 > $OUT/pytest_kmeans_$TAG.log 2>&1; echo "pytest rc=$?"; tail -25 $OUT/pytest_kmeans_$TAG.log | cut -c1-220
echo "== seeding times, 8M x 256, K = 1024, init = k-means++" | tee $OUT/kmpp_times_$TAG.log
run() { echo "## $1" | tee -a $OUT/kmpp_times_$TAG.log; shift; ( "$@" ) 2>&1 | tr '\r' '\n' | grep -E "kmeans_cuda wall|\[timing\] (seeding|set-up)|host chooser" | tee -a $OUT/kmpp_times_$TAG.log; }
run "L2, one shard" env KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --init k-means++ --tolerance 0.5 --yinyang 0 --verbosity 2
run "angular (unit rows, uniform cloud), one shard" env KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --init k-means++ --metric cos --tolerance 0.5 --yinyang 0 --verbosity 2
run "angular, mixture of 1024 Gaussians scaled to unit length, one shard" env KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --init k-means++ --metric cos --data gaussian --tolerance 0.5 --yinyang 0 --verbosity 2
run "config C shape (fp16 angular, 8 virtual shards)" env KMCUDA_AMD_TIMING=1 KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --init k-means++ --metric cos --dtype f16 --yinyang 0.1 --verbosity 2

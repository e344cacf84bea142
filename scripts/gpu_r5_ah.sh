#!/bin/bash
# Busy / idle summaries (rocprofv3 kernel traces) of whole calls: config A (100k x 256 @ 1024, k-means++), k-means++ at 8M
# rows (K = 1024, one Lloyd iteration behind it), knn_cuda for config D's share.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5ah}
python scripts/config_b.py --samples 200000 --verbosity 0 > /dev/null 2>&1
prof() { name=$1; shift; rm -rf $OUT/prof_$TAG; timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_$TAG -o p -- "$@" > $OUT/prof_$TAG.log 2>&1; echo "## $name" | tee -a $OUT/busy_idle_$TAG.log; grep -E "wall|knn_cuda" $OUT/prof_$TAG.log | tee -a $OUT/busy_idle_$TAG.log; python scripts/kernel_timeline.py $OUT/prof_$TAG/p_results.db summary | tee -a $OUT/busy_idle_$TAG.log; rm -rf $OUT/prof_$TAG; }
prof "config A: 100k x 256 @ 1024, k-means++, tolerance 0.01" python scripts/config_b.py --samples 100000 --init k-means++ --verbosity 0
prof "8M x 256 @ 1024, k-means++, tolerance 0.5 (the seeding + two iterations)" python scripts/config_b.py --init k-means++ --tolerance 0.5 --yinyang 0 --verbosity 0
prof "config D share: knn_cuda, 1/8 of 8M queries" python scripts/config_d.py --samples 8000000 --shard 0/8

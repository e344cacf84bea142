#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4i}
echo "== k-NN tests"
timeout 900 python -m pytest tests/test_gpu_knn.py tests/test_gpu_scale.py -k "knn or config_d" -m gpu -q > $OUT/pytest_knn_$TAG.log 2>&1; echo "rc=$?"; tail -4 $OUT/pytest_knn_$TAG.log
CMD="python scripts/config_d.py --samples 8000000 --shard 0/8 --check 100"
KMCUDA_AMD_KNN_STATS=1 timeout 300 $CMD 2>&1 | grep -E "knn_cuda|k-NN filter|brute" | tee $OUT/configD_$TAG.log
echo "== kernel trace"
rm -rf $OUT/profD_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profD_$TAG -o p -- python scripts/config_d.py --samples 8000000 --shard 0/8 > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/profD_$TAG/p_results.db $OUT/kernel_stats_configD_$TAG.csv | head -9 | cut -c1-150
rm -rf $OUT/profD_$TAG

#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4j}
timeout 600 python -m pytest tests/test_gpu_knn.py tests/test_gpu_carry.py -m gpu -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_$TAG.log
KMCUDA_AMD_KNN_STATS=1 timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 --check 200 2>&1 | grep -E "knn_cuda|brute" | tee $OUT/configD_$TAG.log
rm -rf $OUT/profD_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profD_$TAG -o p -- python scripts/config_d.py --samples 8000000 --shard 0/8 > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/profD_$TAG/p_results.db $OUT/kernel_stats_configD_$TAG.csv | head -7 | cut -c1-60,150-230
rm -rf $OUT/profD_$TAG

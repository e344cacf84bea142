#!/bin/bash
# The streamed filter with its chunks staged early (rows behind the first k-step, panel behind the second): parity tests,
# the 2M x 1024 @ 1024 iteration with its kernel stats, the other wide shapes and the 257..512-feature ones.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5z}
timeout 900 python -m pytest tests/test_gpu_wide.py tests/test_gpu_carry.py -m gpu -q -x > $OUT/pytest_wide_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_wide_$TAG.log
for shape in "2000000 1024 100000" "1000000 768 50000" "1000000 1536 50000" "4000000 512 100000" "4000000 384 100000" "4000000 320 100000"; do set -- $shape
timeout 300 python bench.py --samples $1 --features $2 --steps 10 --warmup 5 --no-cpu-baseline --no-api-leg --verify-rows $3 > $OUT/bench_wide_${1}x${2}_$TAG.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/bench_wide_${1}x${2}_$TAG.json').read().strip().splitlines()[-1])
print('$1 x $2', {k:d[k] for k in ('value','ms_per_step')}, d['breakdown_ms_per_step'], d.get('verify',{}).get('ok'))" | tee -a $OUT/wide_shapes_$TAG.log
done
rm -rf $OUT/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o p -- python bench.py --samples 2000000 --features 1024 --steps 10 --warmup 5 --no-cpu-baseline --no-verify --no-api-leg > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_$TAG/p_results.db $OUT/kernel_stats_wide_$TAG.csv | head -8 | awk -F'",' '{print substr($1,1,70), $2}'
rm -rf $OUT/prof_$TAG

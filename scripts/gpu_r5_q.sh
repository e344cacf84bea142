#!/bin/bash
# Wide rows: the contender stage without block barriers (every wave on its own), 3 waves per SIMD without scratch (the
# built library) against 4 with 48 bytes of scratch (scratch/libKMCUDA_wide_cont_cap4.so).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5q}
timeout 600 python -m pytest tests/test_gpu_wide.py -m gpu -q -x > $OUT/pytest_wide_$TAG.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest_wide_$TAG.log
for lib in "" scratch/libKMCUDA_wide_cont_cap4.so "" scratch/libKMCUDA_wide_cont_cap4.so; do
KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python bench.py --samples 2000000 --features 1024 --steps 10 --warmup 5 --no-cpu-baseline --no-api-leg --verify-rows 100000 > $OUT/bench_wide_$TAG.json 2> $OUT/bench_wide_$TAG.err
python3 -c "
import json
d=json.loads(open('$OUT/bench_wide_$TAG.json').read().strip().splitlines()[-1])
print('lib=${lib:-built}', {k:d[k] for k in ('value','ms_per_step')}, d['breakdown_ms_per_step'], d.get('verify',{}).get('ok'))"
done
rm -rf $OUT/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o p -- python bench.py --samples 2000000 --features 1024 --steps 10 --warmup 5 --no-cpu-baseline --no-verify --no-api-leg > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_$TAG/p_results.db $OUT/kernel_stats_wide_$TAG.csv | head -6 | awk -F'",' '{print substr($1,1,70), $2}'
rm -rf $OUT/prof_$TAG
for shape in "1000000 768" "1000000 1536"; do set -- $shape
timeout 300 python bench.py --samples $1 --features $2 --steps 10 --warmup 5 --no-cpu-baseline --no-api-leg --verify-rows 50000 > $OUT/bench_wide_${1}x${2}_$TAG.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/bench_wide_${1}x${2}_$TAG.json').read().strip().splitlines()[-1])
print('$1 x $2', {k:d[k] for k in ('value','ms_per_step')}, d['breakdown_ms_per_step'], d.get('verify',{}).get('ok'))"
done

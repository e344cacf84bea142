#!/bin/bash
# Round 5, third run: the hand-written D > 512 filter (lloyd_wide.hip) -- parity tests, then 2M x 1024 @ 1024 with its
# kernel trace (round 3/4 with rocBLAS: 9.88 ms per iteration).   bash scripts/gpu_r5_c.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5c}
timeout 600 python -m pytest tests/test_gpu_wide.py -m gpu -q -x --durations=5 > $OUT/pytest_wide_$TAG.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest_wide_$TAG.log
echo "== 2M x 1024 @ 1024"
timeout 300 python bench.py --samples 2000000 --features 1024 --steps 10 --warmup 5 --no-cpu-baseline --verify-rows 100000 > $OUT/bench_wide_$TAG.json 2> $OUT/bench_wide_$TAG.err; echo "rc=$?"; tail -3 $OUT/bench_wide_$TAG.err
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_wide_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"), d["rows_full_exact_scan_last_step"], d["rows_pair_refined_last_step"])
PY
rm -rf $OUT/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o p -- python bench.py --samples 2000000 --features 1024 --steps 10 --warmup 5 --no-cpu-baseline --no-verify > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/prof_$TAG/p_results.db $OUT/kernel_stats_wide_$TAG.csv | head -10 | cut -c1-160
rm -rf $OUT/prof_$TAG
for shape in "1000000 768" "1000000 1536"; do set -- $shape
timeout 300 python bench.py --samples $1 --features $2 --steps 10 --warmup 5 --no-cpu-baseline --verify-rows 50000 > $OUT/bench_wide_${1}x${2}_$TAG.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/bench_wide_${1}x${2}_$TAG.json').read().strip().splitlines()[-1])
print('$1 x $2', {k:d[k] for k in ('value','ms_per_step')}, d['breakdown_ms_per_step'], d.get('verify',{}).get('ok'))"
done

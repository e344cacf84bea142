#!/bin/bash
# GPU session D: the tests behind the first failure of session C, then the round's bench evidence:
# bench line, rocprofv3 kernel statistics of the same command, PMC passes, the 1M-row shard, --api.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r2d}
timeout 1200 python -m pytest tests/test_gpu_scale.py tests/test_gpu_sharded.py tests/test_gpu_yinyang.py tests/test_gpu_knn.py -x -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_$TAG.log
echo "== bench"
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; cat $OUT/bench_$TAG.json | head -c 1500; echo
echo "== rocprofv3 kernel trace of the same command"
rm -rf $OUT/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python bench.py --no-cpu-baseline --no-verify > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
f=$(find $OUT/prof_$TAG -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/kernel_stats_$TAG.csv
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:16]:
    print("   %-62s calls %4s avg %9.3f us min %9.3f" % (r["Name"].split("kmx::")[-1][:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
rm -rf $OUT/prof_$TAG
echo "== 1M-row shard"
timeout 300 python bench.py --samples 1000000 --steps 20 --warmup 10 --no-cpu-baseline > $OUT/bench1m_$TAG.json 2> $OUT/bench1m_$TAG.err; echo "rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench1m_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY
echo "== --api (whole kmeans_cuda calls)"
timeout 600 python bench.py --api --steps 20 > $OUT/bench_api_$TAG.json 2> $OUT/bench_api_$TAG.err; echo "rc=$?"; head -c 1200 $OUT/bench_api_$TAG.json; echo
echo "== --gpus 2 on one device over gloo (self-launching path)"
KMCUDA_AMD_BENCH_SINGLE_DEVICE=1 KMCUDA_AMD_BENCH_BACKEND=gloo timeout 300 python bench.py --gpus 2 --steps 5 --warmup 3 --samples 2000000 --no-cpu-baseline > $OUT/bench2_$TAG.json 2> $OUT/bench2_$TAG.err; echo "rc=$?"; head -c 600 $OUT/bench2_$TAG.json; echo; tail -2 $OUT/bench2_$TAG.err
echo "== PMC"
bash scripts/gpu_pmc_all.sh $TAG 2>&1 | tail -12

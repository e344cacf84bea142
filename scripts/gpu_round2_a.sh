#!/bin/bash
# GPU session A of round 2: quick parity subset, the bench at 8M / 1M rows, the stage-1 kernel's
# variants side by side, its phase trace, and the matrix pipe's power probe.  Outputs -> gpurun_out/.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r2a}
echo "== parity subset"
timeout 600 python -m pytest tests/test_gpu_lloyd.py tests/test_gpu_kmeans.py tests/test_gpu_row_cache.py tests/test_gpu_golden.py tests/test_gpu_sharded.py tests/test_gpu_exact_update.py -x -q > $OUT/pytest_$TAG.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_$TAG.log
echo "== bench 8M"
timeout 300 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "rc=$?"; python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench_$TAG.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["breakdown_ms_per_step"], d.get("verify"))
except Exception as e: print("bench parse", e)
PY
tail -3 $OUT/bench_$TAG.err
echo "== bench 1M shard"
timeout 300 python bench.py --samples 1000000 --steps 20 --warmup 10 --no-cpu-baseline > $OUT/bench1m_$TAG.json 2> $OUT/bench1m_$TAG.err; echo "rc=$?"
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench1m_$TAG.json").read().strip().splitlines()[-1])
    print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
except Exception as e: print("bench parse", e)
PY
KMCUDA_AMD_UPDATE=sync timeout 300 python bench.py --samples 1000000 --steps 20 --warmup 10 --no-cpu-baseline --no-verify > $OUT/bench1m_sync_$TAG.json 2>> $OUT/bench1m_$TAG.err
python - <<PY
import json
try:
    d=json.loads(open("$OUT/bench1m_sync_$TAG.json").read().strip().splitlines()[-1])
    print("sync-mode update:", d["ms_per_step"], d["breakdown_ms_per_step"])
except Exception as e: print("bench parse", e)
PY
echo "== stage-1 variants (8M, bench.py)"
for v in base book1 biaspf prio1; do
  KMCUDA_AMD_LIB=scratch/libs/lib$v.so timeout 300 python bench.py --no-cpu-baseline --no-verify --steps 10 --warmup 5 > $OUT/var_${v}_$TAG.json 2> $OUT/var_${v}_$TAG.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/var_${v}_$TAG.json").read().strip().splitlines()[-1])
    print("$v", "ms/step %.3f" % d["ms_per_step"], "coarse %.3f" % d["roofline"]["kernel_ms"], "frac %.3f" % d["roofline"]["frac"])
except Exception as e: print("$v parse", e)
PY
done
echo "== trace"
KMCUDA_AMD_LIB=scratch/libs/libtrace.so timeout 300 python scripts/coarse_trace.py > $OUT/trace_$TAG.log 2>&1; echo "rc=$?"; head -8 $OUT/trace_$TAG.log
echo "== mfma probe"
timeout 300 scratch/bin/mfma_probe 40000 > $OUT/mfma_probe_$TAG.log 2>&1; echo "rc=$?"; cat $OUT/mfma_probe_$TAG.log

#!/bin/bash
# Round 3, session G: stage 2's contender phase over (row, contender) pairs -- parity tests, then A/B against
# the per-lane version (variant library) on the same box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3g}
timeout 1200 python -m pytest -q -x -m gpu tests/test_gpu_lloyd.py tests/test_gpu_row_cache.py tests/test_gpu_golden.py tests/test_gpu_kmeans.py -k "not afkmc2" > $OUT/pytest_${TAG}.log 2>&1
echo "pytest rc=$?"; tail -6 $OUT/pytest_${TAG}.log
for rep in 1 2; do
for lib in default oldsettle; do
  if [ $lib = default ]; then unset KMCUDA_AMD_LIB; else export KMCUDA_AMD_LIB=$PWD/kmcuda_amd/libKMCUDA_$lib.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --verify-rows 300000 > $OUT/bench_${TAG}_8M_${lib}_$rep.json 2>> $OUT/bench_${TAG}.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_${TAG}_8M_${lib}_$rep.json"))
r=d["roofline"]
print("8M $lib $rep", round(d["ms_per_step"],4), "coarse", round(r["kernel_ms"],4), "refine", round(r["filter_stage_ms"]-r["kernel_ms"],4), d["breakdown_ms_per_step"], d["verify"]["ok"])
PY
done; done
for lib in default oldsettle; do
  if [ $lib = default ]; then unset KMCUDA_AMD_LIB; else export KMCUDA_AMD_LIB=$PWD/kmcuda_amd/libKMCUDA_$lib.so; fi
  timeout 200 python bench.py --samples 1000000 --steps 40 --warmup 10 --no-cpu-baseline --verify-rows 200000 > $OUT/bench_${TAG}_1M_$lib.json 2>>$OUT/bench_${TAG}.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_${TAG}_1M_$lib.json"))
r=d["roofline"]
print("1M $lib", round(d["ms_per_step"],4), "coarse", round(r["kernel_ms"],4), "refine", round(r["filter_stage_ms"]-r["kernel_ms"],4), d["verify"]["ok"])
PY
done

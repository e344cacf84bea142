#!/bin/bash
# Round 3, session D: scale parity of configs C and D as tests; the instrumentation-free kernels; the update's
# path steering; A/B of the coarse kernel's DMA placement (variant library, same box).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3d}
timeout 1500 python -m pytest -q -s -m gpu tests/test_gpu_scale.py -k "config_c or config_d" > $OUT/pytest_${TAG}_scale.log 2>&1
echo "scale pytest rc=$?"; grep -E "bounds after|filter pass|answered|brute force|passed|failed|Error|error" $OUT/pytest_${TAG}_scale.log | tail -12
timeout 1200 python -m pytest -q -m gpu tests/test_gpu_lloyd.py tests/test_gpu_yinyang.py tests/test_gpu_kmeans.py -x > $OUT/pytest_${TAG}.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/pytest_${TAG}.log
for rep in 1 2; do
for lib in default dmabook; do
  if [ $lib = default ]; then unset KMCUDA_AMD_LIB; else export KMCUDA_AMD_LIB=$PWD/kmcuda_amd/libKMCUDA_$lib.so; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify > $OUT/bench_${TAG}_8M_${lib}_$rep.json 2>> $OUT/bench_${TAG}.err
  python - <<PY
import json
d=json.load(open("$OUT/bench_${TAG}_8M_${lib}_$rep.json"))
print("$lib $rep", d["ms_per_step"], d["breakdown_ms_per_step"], d["roofline"]["frac"])
PY
done; done
unset KMCUDA_AMD_LIB
echo "== api 8M 1 shard"
timeout 200 python bench.py --api --steps 20 > $OUT/bench_${TAG}_api1.json 2>>$OUT/bench_${TAG}.err
python -c "import json;d=json.load(open('$OUT/bench_${TAG}_api1.json'));print(d['ms_per_step'], [c['loop_s'] for c in d['calls']])"

#!/bin/bash
# A/B of two builds of the library on ONE box by rocprofv3 kernel traces of the bench (8M rows and the 1M-row
# shard): per kernel (launches, mean of the last 10 launches in us, minimum in us).
#   bash scripts/gpu_ab_trace.sh <tag> <variant>      variant: kmcuda_amd/libKMCUDA_<variant>.so
# PYTEST="tests/a.py tests/b.py" runs those GPU tests against the default build first.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-ab}; VAR=${2:-variant}
if [ -n "${PYTEST:-}" ]; then
  timeout 1500 python -m pytest -q -x -m gpu $PYTEST > $OUT/pytest_${TAG}.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest_${TAG}.log
fi
for lib in default $VAR; do
  if [ $lib = default ]; then unset KMCUDA_AMD_LIB; else export KMCUDA_AMD_LIB=$PWD/kmcuda_amd/libKMCUDA_$lib.so; fi
  for n in 8000000 1000000; do
    rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_${lib}_$n -o p -- python bench.py --samples $n --steps 10 --warmup 10 --no-cpu-baseline --no-verify > $OUT/bench_${TAG}_${lib}_$n.json 2>/dev/null
    python - <<PY
import sqlite3, json, re
db = sqlite3.connect("$OUT/prof_${TAG}_${lib}_$n/p_results.db")
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
per = {}
for nm, a, b in cur.execute("select %s, start, end from kernels order by start" % name):
    if "kmx" not in nm:
        continue
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", nm)
    key = (m.group(1) + (m.group(2) or ""))[:44] if m else nm[:44]
    per.setdefault(key, []).append(b - a)
d = json.load(open("$OUT/bench_${TAG}_${lib}_$n.json"))
keep = ("coarse2", "refine", "settle", "cluster_sums", "scatter", "prep_frozen")
out = {k: (len(v), round(sum(v[-10:]) / len(v[-10:]) / 1e3, 1), round(min(v) / 1e3, 1)) for k, v in per.items() if any(x in k for x in keep)}
print("$lib", $n, "ms/step", round(d["ms_per_step"], 4), out)
PY
    rm -rf $OUT/prof_${TAG}_${lib}_$n
  done
done

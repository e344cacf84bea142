#!/bin/bash
# Round 3, session I: cluster_sums with 16-byte loads -- update tests, then A/B by kernel trace against the
# scalar mapping (variant library) on one box.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3i}
for lib in default sums4b; do
  if [ $lib = default ]; then unset KMCUDA_AMD_LIB; else export KMCUDA_AMD_LIB=$PWD/kmcuda_amd/libKMCUDA_$lib.so; fi
  for n in 8000000 1000000; do
    rocprofv3 --kernel-trace --stats -d $OUT/prof_${TAG}_${lib}_$n -o p -- python bench.py --samples $n --steps 10 --warmup 10 --no-cpu-baseline --no-verify > $OUT/bench_${TAG}_${lib}_$n.json 2>/dev/null
    python - <<PY
import sqlite3, json
db = sqlite3.connect("$OUT/prof_${TAG}_${lib}_$n/p_results.db")
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = list(cur.execute("select %s, start, end from kernels order by start" % name))
per = {}
for nm, a, b in rows:
    if "kmx::" in nm or "_ZN3kmx" in nm:
        import re
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", nm)
        key = (m.group(1) + (m.group(2) or ""))[:48] if m else nm[:48]
        per.setdefault(key, []).append(b - a)
d = json.load(open("$OUT/bench_${TAG}_${lib}_$n.json"))
out = {k: (len(v), round(sum(v[-10:]) / len(v[-10:]) / 1e3, 1), round(min(v) / 1e3, 1)) for k, v in per.items() if "cluster_sums" in k or "scatter" in k}
print("$lib", $n, "ms/step", round(d["ms_per_step"], 4), out)
PY
    rm -rf $OUT/prof_${TAG}_${lib}_$n
  done
done

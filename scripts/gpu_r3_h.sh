#!/bin/bash
# Round 3, session H: stage 2 with and without the scratch (variant library = the kernel of commit 57838d5),
# per-kernel durations from rocprofv3 kernel traces on ONE box (the HIP-event split of the filter span into
# stage 1 / stage 2 differs from box to box).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r3h}
for lib in default refinespill; do
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
        key = nm.split("kmx::")[-1].split("(")[0][:40] if "kmx::" in nm else nm[7:40]
        per.setdefault(key, []).append(b - a)
d = json.load(open("$OUT/bench_${TAG}_${lib}_$n.json"))
out = {k: (len(v), round(sum(v[-10:]) / len(v[-10:]) / 1e3, 1), round(min(v) / 1e3, 1)) for k, v in per.items() if "refine" in k or "coarse2" in k or "settle" in k or "cluster_sums<true>" in k or "prep_frozen" in k}
print("$lib", $n, "ms/step", round(d["ms_per_step"], 4), out)
PY
    rm -rf $OUT/prof_${TAG}_${lib}_$n
  done
done

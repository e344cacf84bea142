#!/bin/bash
# GPU session C: the whole -m gpu suite, then BASELINE configs B, C, D at their named per-GPU sizes with
# rocprofv3 kernel statistics.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r2c}
echo "== pytest -m gpu"
timeout 1700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_gpu_$TAG.log
echo "== config B as named (Yinyang 0.1), kernel stats"
rm -rf /tmp/pb
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o b -- python scripts/config_b.py --yinyang 0.1 --verbosity 0 > $OUT/config_b_$TAG.log 2>&1; echo "rc=$?"
grep -E "wall|clusters" $OUT/config_b_$TAG.log
f=$(find /tmp/pb -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/config_b_kernel_stats_$TAG.csv 2>/dev/null
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("   %-60s calls %4s total %9.2f ms avg %9.3f ms" % (r["Name"].split("kmx::")[-1][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
echo "== config B, Lloyd (yinyang 0), un-profiled"
timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0 2>&1 | grep wall
timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0 2>&1 | grep wall
echo "== config C: 8M x 256 fp16 angular Yinyang, 8 row shards (virtual: one GPU runs them in turn)"
KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 900 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0 > $OUT/config_c_$TAG.log 2>&1; echo "rc=$?"; grep -E "wall|clusters" $OUT/config_c_$TAG.log
timeout 300 python scripts/config_b.py --samples 1000000 --metric cos --dtype f16 --yinyang 0.1 --verbosity 0 2>&1 | grep wall
echo "== config D: 8M x 256 corpus, the queries of rank 0 of 8"
rm -rf /tmp/pd
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pd -o d -- python scripts/config_d.py --samples 8000000 --shard 0/8 --check 20 > $OUT/config_d_$TAG.log 2>&1; echo "rc=$?"
grep -E "knn_cuda|calculated|brute" $OUT/config_d_$TAG.log
f=$(find /tmp/pd -name '*kernel_stats.csv' | head -1); cp "$f" $OUT/config_d_kernel_stats_$TAG.csv 2>/dev/null
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print("   %-60s calls %4s total %9.2f ms avg %9.3f ms" % (r["Name"].split("kmx::")[-1][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY

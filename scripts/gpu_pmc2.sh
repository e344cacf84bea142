#!/bin/bash
# SQ counters of the assignment filter kernels (one pass, kernel-trace only)
set -u
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-r1}
mkdir -p $OUT
rm -rf /tmp/pmc2
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc2 -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc2_$TAG.log 2>&1
f=$(find /tmp/pmc2 -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "lloyd_filter" in n or "lloyd_coarse" in n:
        key = "coarse" if "coarse" in n else ("f16" if "f16" in n else "f32")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    wc = sum(v["SQ_WAVE_CYCLES"]) / len(v["SQ_WAVE_CYCLES"])
    for c, vals in sorted(v.items()):
        m = sum(vals) / len(vals)
        print(k, c, "n=%d mean=%.4g  (%.1f%% of wave cycles)" % (len(vals), m, 100 * m / wc))
PY

#!/bin/bash
# Where the hand-over point's 6 ms (4M rows) go: timing laps + a kernel trace of the short mixture call, both schedules.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5g}
for yy in 0.1 0; do
  echo "## yinyang_t=$yy" | tee -a $OUT/handover_$TAG.log
  KMCUDA_AMD_TIMING=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang $yy --tolerance 0.01 --verbosity 0 2>&1 | grep -E "timing|wall" | tee -a $OUT/handover_$TAG.log
done
for yy in 0.1 0; do
  rm -rf $OUT/prof_$TAG
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_$TAG -o p -- python scripts/config_b.py --samples 4000000 --data gaussian --yinyang $yy --tolerance 0.01 --verbosity 0 > /dev/null 2>&1
  python3 - $yy <<'PY' | tee -a $OUT/handover_$TAG.log
import csv, glob, sys
rows = []
for f in glob.glob("gpurun_out/prof_*/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "kmx::" in r["Kernel_Name"] or "rocprim" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
print("## kernel timeline, yinyang_t=%s (ms from the first kernel: start, duration, name)" % sys.argv[1])
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    if d > 0.02:
        n = r["Kernel_Name"].split("kmx::")[-1].split("(")[0][:60]
        print("%9.3f %8.3f %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, d, n))
PY
  rm -rf $OUT/prof_$TAG
done

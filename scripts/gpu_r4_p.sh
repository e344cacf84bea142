#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
rm -rf /tmp/hiptrace
KMCUDA_AMD_PRELOAD=0 timeout 300 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d /tmp/hiptrace -o t -- python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --verbosity 0 > /dev/null 2>&1
ls /tmp/hiptrace/*/ | head
python3 - <<'PY' | tee $OUT/hip_api_first_iteration.log
import csv, glob
api = glob.glob("/tmp/hiptrace/**/*hip_api_trace.csv", recursive=True)
ker = glob.glob("/tmp/hiptrace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(api[0])))
k = list(csv.DictReader(open(ker[0])))
# the window: from the first lloyd_settle kernel's end to the first move_count kernel's start
se = [int(r["End_Timestamp"]) for r in k if "lloyd_settle" in r["Kernel_Name"]]
ms = [int(r["Start_Timestamp"]) for r in k if "move_count" in r["Kernel_Name"]]
a, b = min(se), min(ms)
print("gap between the first settle kernel's end and the first move_count kernel's start: %.3f ms" % ((b - a) / 1e6))
calls = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in rows]
calls.sort()
print("HIP API calls that overlap the window, longer than 0.2 ms:")
for s, e, f in calls:
    if e > a - 3e6 and s < b + 1e6 and (e - s) > 2e5:
        print("  %-40s start %+9.3f ms (rel. settle end)  duration %8.3f ms" % (f, (s - a) / 1e6, (e - s) / 1e6))
print("longest HIP API calls of the whole run:")
for s, e, f in sorted(calls, key=lambda c: c[0] - c[1])[:12]:
    print("  %-40s %8.3f ms" % (f, (e - s) / 1e6))
PY

#!/usr/bin/env python
"""Per-kernel summary of rocprofv3 PMC passes (one run per counter group, scripts/gpu.sh pmc): per-launch means over
each kernel's last PMC_LAST launches, FETCH_SIZE corrected as MI355X_MICROARCH.md's HBM section prescribes for gfx950.
usage: PMC_TAG=<tag> python scripts/pmc_summary.py <out.json>   (reads /tmp/pmc_*/**/*counter_collection.csv)"""

import csv, sys, glob, json, collections, os
LAST = int(os.environ.get("PMC_LAST", "10"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
names = {}
for f in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        n = r["Kernel_Name"]
        if "kmx::" not in n:
            continue
        key = n.split("kmx::")[1].split("(")[0]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
            dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"source": "scripts/gpu.sh pmc: rocprofv3 --pmc <group> --kernel-trace, one run per group, "
                 "python bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-verify --no-api-leg; per-launch means over each kernel's LAST 10 launches (the timed iterations)",
       "units": "FETCH_SIZE/WRITE_SIZE in KB as reported; fetch_bytes_corrected = 2 x FETCH_SIZE x 1024 (gfx950: "
                "wide streaming reads are tallied at half, MI355X_MICROARCH.md HBM); SQ_* summed over SIMDs "
                "(quad-cycles for *_CYCLES waits per the guide); GRBM_GUI_ACTIVE summed over 8 XCDs",
       "rows_per_launch": 8000000, "collected": __import__("time").strftime("%Y-%m-%dT%H:%M:%S"), "tag": os.environ.get("PMC_TAG", ""),
       "kernels": {}}
for k, v in sorted(agg.items()):
    e = {c: sum(x[-LAST:]) / len(x[-LAST:]) for c, x in v.items()}
    e["launches"] = max(len(x) for x in v.values())
    e["launches_averaged"] = min(LAST, e["launches"])
    if "FETCH_SIZE" in e:
        e["fetch_bytes_corrected"] = 2.0 * e["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in e:
        e["write_bytes"] = e["WRITE_SIZE"] * 1024.0
    if "fetch_bytes_corrected" in e and "write_bytes" in e:
        e["traffic_bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
    if dur[k] and "GRBM_GUI_ACTIVE" in e:
        d = sum(dur[k][-LAST:]) / len(dur[k][-LAST:])
        e["launch_ms_under_pmc"] = d / 1e6
        e["effective_clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / d
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e:
            e["mfma_busy_fraction_of_active_cycles"] = (e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (e["GRBM_GUI_ACTIVE"] / 8.0)
    out["kernels"][k] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k in out["kernels"]:
    if "lloyd" in k or "knn_filter" in k:
        print(k, {a: (round(b, 4) if b < 100 else float("%.4g" % b)) for a, b in out["kernels"][k].items()})

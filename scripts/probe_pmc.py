#!/usr/bin/env python
"""scripts/probe_pmc.py <rocprofv3 pmc csv dir> -- per kernel launch of a standalone probe (scripts/coarse_probe.hip,
scripts/mfma_probe.hip) run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace
--output-format csv`: duration, the clock the chip held (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the share of those
cycles the matrix pipe was busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) -- the formulas of scripts/pmc_summary.py.
What it shows: with the pipe saturated on random halves the chip clocks DOWN, and throughput = clock x busy."""
import csv
import glob
import sys
from collections import OrderedDict

rows = []
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
launch = OrderedDict()
for r in rows:
    key = int(r["Dispatch_Id"])
    e = launch.setdefault(key, {"name": r["Kernel_Name"].split("(")[0][-60:], "t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("%-62s %9s %9s %9s" % ("kernel (launch order)", "ms", "GHz", "MFMA busy"))
for k, e in sorted(launch.items()):
    if "GRBM_GUI_ACTIVE" not in e or e["t"] <= 0:
        continue
    clk = e["GRBM_GUI_ACTIVE"] / 8.0 / e["t"]
    busy = (e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0) / (e["GRBM_GUI_ACTIVE"] / 8.0)
    print("%-62s %9.3f %9.3f %9.3f" % (e["name"], e["t"] / 1e6, clk, busy))

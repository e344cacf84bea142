export TMPDIR=/tmp
rm -rf /tmp/pmc3
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc3 -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /tmp/pmc3.log 2>&1
f=$(find /tmp/pmc3 -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "lloyd_filter" in n or "lloyd_coarse" in n:
        key = "coarse" if "coarse" in n else ("f16" if "f16" in n else "f32")
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    d = sum(dur[k]) / len(dur[k])
    g = sum(v["GRBM_GUI_ACTIVE"]) / len(v["GRBM_GUI_ACTIVE"])
    m = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"])
    print(k, "duration ms %.3f  GUI_ACTIVE/8 = %.4g cycles -> clock %.3f GHz; MFMA busy per SIMD %.4g cycles = %.1f%% of GUI" % (d / 1e6, g / 8, g / 8 / d, m / 1024, 100 * (m / 1024) / (g / 8)))
PY

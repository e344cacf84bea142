#!/bin/bash
# PMC passes over the k-NN filter kernel (config D shape at 2M x 256, K = 1024, share 0 of 2) -> gpurun_out/pmc_knn_<tag>.json
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; TAG=${1:-r2}; mkdir -p $OUT
CMD="python scripts/config_d.py --samples 2000000 --shard 0/2"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pk_$i -o pmc -- $CMD > /tmp/pk_$i.log 2>&1
  echo "pmc pass $i ($grp) rc=$?"; grep -E "knn_cuda|calculated" /tmp/pk_$i.log | head -2
done
python3 - "$OUT/pmc_knn_${TAG}.json" <<'PY'
import csv, sys, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pk_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "kmx::knn" not in n: continue
        key = n.split("kmx::")[1].split("(")[0]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"source": "scripts/gpu_pmc_knn.sh: rocprofv3 --pmc <group> --kernel-trace, one run per group, python scripts/config_d.py --samples 2000000 --shard 0/2", "kernels": {}}
for k, v in sorted(agg.items()):
    e = {c: sum(x) / len(x) for c, x in v.items()}
    if "FETCH_SIZE" in e: e["fetch_bytes_corrected"] = 2.0 * e["FETCH_SIZE"] * 1024.0
    if "WRITE_SIZE" in e: e["write_bytes"] = e["WRITE_SIZE"] * 1024.0
    if dur[k] and "GRBM_GUI_ACTIVE" in e:
        d = sum(dur[k]) / len(dur[k]); e["launch_ms_under_pmc"] = d / 1e6; e["effective_clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8.0 / d
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e: e["mfma_busy_fraction_of_active_cycles"] = (e["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (e["GRBM_GUI_ACTIVE"] / 8.0)
    out["kernels"][k] = e
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, e in out["kernels"].items():
    if "filter" in k: print(k, {a: (round(b, 4) if b < 100 else float("%.4g" % b)) for a, b in e.items()})
PY

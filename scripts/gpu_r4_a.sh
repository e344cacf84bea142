#!/bin/bash
# Round 4, first session: the tests of the round's first commit (overflow pin, transpose, export list, restartable
# loop), the default bench line, and the evidence the round-3 verdict asked for on the k-NN search of BASELINE
# config D (8M x 256 corpus, the 1M queries of rank 0 of 8): wall clock with the device statistics, a kernel trace,
# four PMC passes.     bash scripts/gpu_r4_a.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4a}
echo "== targeted tests"
timeout 900 python -m pytest tests/test_gpu_scale.py -k "overflow" tests/test_gpu_lloyd.py -k "overflow or transpose or stream" -m gpu -q -x > $OUT/pytest_a_$TAG.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_a_$TAG.log
timeout 600 python -m pytest tests/test_gpu_knn.py tests/test_gpu_sharded.py -m gpu -q -x > $OUT/pytest_b_$TAG.log 2>&1; echo "rc=$?"; tail -3 $OUT/pytest_b_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
echo "== bench"
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"; head -c 1500 $OUT/bench_$TAG.json; echo
echo "== config D share: wall clock + device statistics"
CMD="python scripts/config_d.py --samples 8000000 --shard 0/8"
KMCUDA_AMD_KNN_STATS=1 timeout 300 $CMD 2>&1 | grep -E "knn_cuda|calculated|k-NN filter" | tee $OUT/configD_$TAG.log
KMCUDA_AMD_KNN_STATS=1 KMCUDA_AMD_KNN_TIGHT=0 timeout 300 $CMD 2>&1 | grep -E "knn_cuda|calculated|k-NN filter" | sed 's/^/TIGHT=0: /' | tee -a $OUT/configD_$TAG.log
echo "== config D share: kernel trace"
rm -rf $OUT/profD_$TAG
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profD_$TAG -o p -- $CMD > $OUT/profD_$TAG.log 2>&1; echo "rocprof rc=$?"
python scripts/rocpd_stats.py $OUT/profD_$TAG/p_results.db $OUT/kernel_stats_configD_$TAG.csv | head -12 | cut -c1-170
rm -rf $OUT/profD_$TAG
echo "== config D share: PMC passes"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/pk_$i
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pk_$i -o pmc -- $CMD > /tmp/pk_$i.log 2>&1
  echo "pmc pass $i ($grp) rc=$?"; grep -E "knn_cuda" /tmp/pk_$i.log | head -1
done
python3 - "$OUT/pmc_knn_configD_${TAG}.json" "$CMD" <<'PY'
import csv, sys, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("/tmp/pk_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "kmx::knn" not in n: continue
        key = n.split("kmx::")[1].split("(")[0]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {"source": "scripts/gpu_r4_a.sh: rocprofv3 --pmc <group> --kernel-trace, one run per group, " + sys.argv[2],
       "units": "FETCH_SIZE/WRITE_SIZE in KB as reported; fetch_bytes_corrected = 2 x FETCH_SIZE x 1024 (gfx950, MI355X_MICROARCH.md); "
                "SQ_* summed over SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs", "kernels": {}}
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
    if "filter" in k or "bounds" in k: print(k, {a: (round(b, 4) if b < 100 else float("%.4g" % b)) for a, b in e.items()})
PY

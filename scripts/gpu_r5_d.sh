#!/bin/bash
# Round 5, fourth run: (1) the k-NN dispatch order (one query cluster's blocks per XCD) against the plain order on
# config D's share, with FETCH_SIZE; (2) the wide filter's HBM traffic and the non-temporal row loads of its contender
# stage.   bash scripts/gpu_r5_d.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5d}
timeout 600 python -m pytest tests/test_gpu_knn.py -m gpu -q -x > $OUT/pytest_knn_$TAG.log 2>&1; echo "pytest knn rc=$?"; tail -3 $OUT/pytest_knn_$TAG.log
echo "== config D share, dispatch order A/B" | tee $OUT/knn_xcd_$TAG.log
for x in 1 0 1 0; do
  echo "## KMCUDA_AMD_KNN_XCD=$x" | tee -a $OUT/knn_xcd_$TAG.log
  KMCUDA_AMD_KNN_XCD=$x KMCUDA_AMD_KNN_STATS=1 timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 2>&1 | grep -E "knn_cuda|k-NN filter" | tee -a $OUT/knn_xcd_$TAG.log
done
echo "== FETCH_SIZE of the filter kernel, both orders" | tee -a $OUT/knn_xcd_$TAG.log
for x in 1 0; do
  rm -rf /tmp/pk_$x
  KMCUDA_AMD_KNN_XCD=$x timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pk_$x -o pmc -- python scripts/config_d.py --samples 8000000 --shard 0/8 > /tmp/pk_$x.log 2>&1
  echo "pmc rc=$? (XCD=$x)"; grep -E "knn_cuda" /tmp/pk_$x.log | tee -a $OUT/knn_xcd_$TAG.log
  python3 - $x <<'PY' | tee -a $OUT/knn_xcd_$TAG.log
import csv, glob, sys
for f in glob.glob("/tmp/pk_%s/**/*counter_collection.csv" % sys.argv[1], recursive=True):
    for r in csv.DictReader(open(f)):
        if "knn_filter_f16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            kb = float(r["Counter_Value"]); ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            print("KMCUDA_AMD_KNN_XCD=%s knn_filter_f16_kernel: FETCH_SIZE %.4g KB -> corrected (x2) %.4g TB, %.1f ms under the counter" % (sys.argv[1], kb, 2 * kb * 1024 / 1e12, ms))
PY
done
echo "== wide rows: bench + traffic of lloyd_wide<0>"
timeout 300 python bench.py --samples 2000000 --features 1024 --steps 10 --warmup 5 --no-cpu-baseline --verify-rows 100000 > $OUT/bench_wide_$TAG.json 2> $OUT/bench_wide_$TAG.err; echo "rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_wide_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"))
PY
for grp in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pw_$grp
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/pw_$grp -o pmc -- python bench.py --samples 2000000 --features 1024 --steps 6 --warmup 4 --no-cpu-baseline --no-verify > /tmp/pw_$grp.log 2>&1; echo "pmc $grp rc=$?"
done
python3 - <<'PY' | tee $OUT/pmc_wide_$TAG.log
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pw_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("lloyd_wide_kernelILi0", "lloyd_wide_kernelILi1", "wide_contenders_kernel"):
            if key in n: agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    out = {c: sum(x[-6:]) / len(x[-6:]) for c, x in v.items()}
    fetch = 2 * out.get("FETCH_SIZE", 0) * 1024; wr = out.get("WRITE_SIZE", 0) * 1024
    print(k, "per launch (last 6): fetch (x2 corrected) %.3f GB, write %.3f GB" % (fetch / 1e9, wr / 1e9))
print("operands: 2M x 1024 halves = 4.096 GB (+ 32 MB of records); panel 2 MB")
PY

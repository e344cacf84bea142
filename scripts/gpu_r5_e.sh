#!/bin/bash
# Round 5, fifth run: the carry-or-not rule (tests + the round-4 table's losing rows), the k-NN filter as 8-wave blocks
# (512 queries per staged tile, one block per CU) against the default 4-wave blocks.   bash scripts/gpu_r5_e.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5e}
timeout 900 python -m pytest tests/test_gpu_carry.py -m gpu -q -x > $OUT/pytest_carry_$TAG.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_carry_$TAG.log
echo "== whole calls where the bounds used to lose (DESIGN 4.4's table)" | tee $OUT/configs_carry_$TAG.log
run() { echo "## $1" | tee -a $OUT/configs_carry_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds|more iterations" | tee -a $OUT/configs_carry_$TAG.log; }
for rep in 1 2; do
run "4M-row mixture tol 0.01: default (yinyang_t=0.1)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.01 --verbosity 0
run "4M-row mixture tol 0.01: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.01 --verbosity 0
done
run "4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.0001 --verbosity 0
run "config B: default" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B: yinyang_t=0" timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
echo "== k-NN filter: 4-wave blocks (default build) against 8-wave blocks (scratch/libKMCUDA_knn8w.so)" | tee $OUT/knn_8wave_$TAG.log
for lib in "" scratch/libKMCUDA_knn8w.so "" scratch/libKMCUDA_knn8w.so; do
  echo "## KMCUDA_AMD_LIB=$lib" | tee -a $OUT/knn_8wave_$TAG.log
  KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} KMCUDA_AMD_KNN_STATS=1 timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 --check 64 2>&1 | grep -E "knn_cuda|k-NN filter|brute" | tee -a $OUT/knn_8wave_$TAG.log
done
rm -rf /tmp/pk8
KMCUDA_AMD_LIB=$GRAFT_REPO_ROOT/scratch/libKMCUDA_knn8w.so timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pk8 -o pmc -- python scripts/config_d.py --samples 8000000 --shard 0/8 > /tmp/pk8.log 2>&1
python3 - <<'PY' | tee -a $OUT/knn_8wave_$TAG.log
import csv, glob
for f in glob.glob("/tmp/pk8/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "knn_filter_f16" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            kb = float(r["Counter_Value"]); ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            print("8-wave blocks: knn_filter_f16_kernel FETCH_SIZE %.4g KB -> corrected (x2) %.4g TB, %.1f ms under the counter" % (kb, 2 * kb * 1024 / 1e12, ms))
PY

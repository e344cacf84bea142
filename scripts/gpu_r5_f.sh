#!/bin/bash
# Round 5, sixth run: the delayed start of the carried bounds (tests + the table's rows), the k-NN filter's 8-wave
# blocks as the default build with deeper tile rings as variants, the k-NN tests.   bash scripts/gpu_r5_f.sh <tag>
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5f}
timeout 900 python -m pytest tests/test_gpu_carry.py tests/test_gpu_knn.py -m gpu -q -x > $OUT/pytest_carry_knn_$TAG.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest_carry_knn_$TAG.log
echo "== whole calls (DESIGN 4.4's table)" | tee $OUT/configs_carry_$TAG.log
run() { echo "## $1" | tee -a $OUT/configs_carry_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|carried bounds" | tee -a $OUT/configs_carry_$TAG.log; }
for rep in 1 2; do
run "4M-row mixture tol 0.01: default (yinyang_t=0.1)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.01 --verbosity 0
run "4M-row mixture tol 0.01: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.01 --verbosity 0
run "4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "4M-row mixture tol 1e-4: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0 --tolerance 0.0001 --verbosity 0
done
run "4M-row mixture tol 1e-4: default, verbosity 2 (spared count)" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 2
run "4M-row mixture tol 1e-4: KMCUDA_AMD_CARRY=1 (bounds from the hand-over on, round 4's rule)" env KMCUDA_AMD_CARRY=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "config B: default" timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B: yinyang_t=0" timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
run "angular 4M-row mixture tol 1e-4: default" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0.1 --tolerance 0.0001 --verbosity 0
run "angular 4M-row mixture tol 1e-4: yinyang_t=0" timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --metric cos --yinyang 0 --tolerance 0.0001 --verbosity 0
echo "== k-NN filter (8-wave blocks): tile ring of 2 (default build), 3, 4" | tee $OUT/knn_ring_$TAG.log
for lib in "" scratch/libKMCUDA_knn_nbuf3.so scratch/libKMCUDA_knn_nbuf4.so "" scratch/libKMCUDA_knn_nbuf3.so scratch/libKMCUDA_knn_nbuf4.so; do
  echo "## KMCUDA_AMD_LIB=$lib" | tee -a $OUT/knn_ring_$TAG.log
  KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib} timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 --check 32 2>&1 | grep -E "knn_cuda|brute" | tee -a $OUT/knn_ring_$TAG.log
done
echo "== k-NN 512 features (4 waves, one set, two blocks per CU -- unchanged): 400000 rows" | tee -a $OUT/knn_ring_$TAG.log
timeout 300 python scripts/config_d.py --samples 400000 --features 512 --clusters 256 2>&1 | grep -E "knn_cuda" | tee -a $OUT/knn_ring_$TAG.log

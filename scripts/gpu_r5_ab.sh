#!/bin/bash
# k-NN filter: when the pieces of the next candidate tile are issued (every DSTR-th k-step; the built library spreads them
# over the tile: 6) -- variant builds -DKNN16_DSTR=1 / 2 / 3, config D's share, one box, interleaved.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5ab}
for lib in "" knn_dstr1 knn_dstr2 knn_dstr3 "" knn_dstr1 knn_dstr2 knn_dstr3; do
  echo "## KMCUDA_AMD_LIB=$lib" | tee -a $OUT/knn_dstr_$TAG.log
  KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/scratch/libKMCUDA_$lib.so} KMCUDA_AMD_KNN_STATS=1 timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 --check 32 2>&1 | grep -E "knn_cuda|brute|k-NN filter" | cut -c1-260 | tee -a $OUT/knn_dstr_$TAG.log
done

#!/bin/bash
# A/B of k-NN filter builds in scratch/libs/libknn_*.so against the in-tree library (config D shape)
cd "$GRAFT_REPO_ROOT"
for lib in "" $(ls scratch/libs/libknn_*.so 2>/dev/null); do
  for cfg in "--samples 2000000 --shard 0/2" "--samples 8000000 --shard 0/8"; do
    if [ -n "$lib" ]; then export KMCUDA_AMD_LIB=$GRAFT_REPO_ROOT/$lib; else unset KMCUDA_AMD_LIB; fi
    echo "${lib:-default} $cfg: $(timeout 250 python scripts/config_d.py $cfg $EXTRA 2>&1 | grep knn_cuda | sed 's/.*knn_cuda/knn_cuda/')"
  done
done

#!/bin/bash
# The update's path rule, A/B on one box: the built library (an older report stays valid) against round 4's rule (the
# largest list zeroed behind a radix call: scratch/libKMCUDA_update_zeroing.so), by path counts and loop times.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5n}
run() { echo "## $1" | tee -a $OUT/update_rule_$TAG.log; shift; ( "$@" ) 2>&1 | grep -E "kmeans_cuda wall|\[update\]" | tee -a $OUT/update_rule_$TAG.log; }
for lib in "" scratch/libKMCUDA_update_zeroing.so "" scratch/libKMCUDA_update_zeroing.so; do
export KMCUDA_AMD_LIB=${lib:+$GRAFT_REPO_ROOT/$lib}
[ -z "$lib" ] && unset KMCUDA_AMD_LIB
echo "#### library: ${lib:-the built one}" | tee -a $OUT/update_rule_$TAG.log
run "config C shape (fp16 angular, 8 virtual shards), init random" env KMCUDA_AMD_UPDATE_TRACE=1 KMCUDA_AMD_VIRTUAL_SHARDS=8 timeout 300 python scripts/config_b.py --metric cos --dtype f16 --yinyang 0.1 --verbosity 0
run "config B default" env KMCUDA_AMD_UPDATE_TRACE=1 timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0
run "config B yinyang_t=0" env KMCUDA_AMD_UPDATE_TRACE=1 timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0
run "4M-row mixture tol 0.01 default" env KMCUDA_AMD_UPDATE_TRACE=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.01 --verbosity 0
run "4M-row mixture tol 1e-4 default" env KMCUDA_AMD_UPDATE_TRACE=1 timeout 300 python scripts/config_b.py --samples 4000000 --data gaussian --yinyang 0.1 --tolerance 0.0001 --verbosity 0
done

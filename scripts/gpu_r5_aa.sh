#!/bin/bash
# Stage 2 of the headline step (lloyd_refine_kernel): how much is the sweep over the listed rows, how much the contender
# phase?  Kernel stats of the built library and of a timing-only build whose blocks leave behind the sweep.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5aa}
python bench.py --samples 200000 --steps 2 --warmup 1 --no-cpu-baseline --no-api-leg --no-verify > /dev/null 2>&1
for lib in "" refine_sweep_only; do
if [ -z "$lib" ]; then unset KMCUDA_AMD_LIB; else export KMCUDA_AMD_LIB=$GRAFT_REPO_ROOT/scratch/libKMCUDA_$lib.so; fi
rm -rf $OUT/prof_$TAG
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-verify --no-api-leg > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
echo "## ${lib:-built}" | tee -a $OUT/refine_split_$TAG.log
python scripts/rocpd_stats.py $OUT/prof_$TAG/p_results.db $OUT/kernel_stats_${lib:-built}_$TAG.csv | head -9 | awk -F'",' '{print substr($1,1,60), $2}' | tee -a $OUT/refine_split_$TAG.log
rm -rf $OUT/prof_$TAG
done

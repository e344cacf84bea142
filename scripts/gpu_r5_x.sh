#!/bin/bash
# The streamed filter's staging schedule: which pieces of the next chunk are issued when (variant builds of lloyd_wide.hip,
# -DWIDE_STAGE_ORDER=n, scratch/libKMCUDA_stage<n>.so) -- one box, interleaved.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5x}
python bench.py --samples 200000 --features 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-api-leg --no-verify > /dev/null 2>&1
for round in 1 2; do
for lib in "" 1 2 3 4 5; do
if [ -z "$lib" ]; then unset KMCUDA_AMD_LIB; else export KMCUDA_AMD_LIB=$GRAFT_REPO_ROOT/scratch/libKMCUDA_stage$lib.so; fi
for shape in "2000000 1024" "4000000 384"; do set -- $shape
timeout 300 python bench.py --samples $1 --features $2 --steps 10 --warmup 5 --no-cpu-baseline --no-api-leg --verify-rows 100000 > $OUT/bench_stage_$TAG.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/bench_stage_$TAG.json').read().strip().splitlines()[-1])
print('order ${lib:-0}: $1 x $2', d['ms_per_step'], d['breakdown_ms_per_step'], d.get('verify',{}).get('ok'))" | tee -a $OUT/stage_order_$TAG.log
done
done
done

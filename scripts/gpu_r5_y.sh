#!/bin/bash
# What bounds the streamed filter's stage 1: timing-only builds (wrong results) -- every block streaming the same 256 rows
# (L2 hits), and nothing staged behind the prologue (products + fragment reads only) -- beside the real kernel.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5y}
python bench.py --samples 200000 --features 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-api-leg --no-verify > /dev/null 2>&1
for lib in stage3 whatif1 whatif2 stage3 whatif1 whatif2; do
export KMCUDA_AMD_LIB=$GRAFT_REPO_ROOT/scratch/libKMCUDA_$lib.so
for shape in "2000000 1024" "4000000 384"; do set -- $shape
timeout 300 python bench.py --samples $1 --features $2 --steps 10 --warmup 5 --no-cpu-baseline --no-api-leg --no-verify > $OUT/bench_whatif_$TAG.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/bench_whatif_$TAG.json').read().strip().splitlines()[-1])
print('$lib: $1 x $2', d['ms_per_step'], d['breakdown_ms_per_step'])" | tee -a $OUT/whatif_$TAG.log
done
done

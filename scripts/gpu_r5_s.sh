#!/bin/bash
# Rows of 257..512 features: the register-resident one-operand-set filter against the LDS-streamed one (KMCUDA_AMD_WIDE_MIN_D).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5s}
for shape in "4000000 512" "4000000 384" "4000000 320"; do set -- $shape
for wide in "" 257; do
KMCUDA_AMD_WIDE_MIN_D=$wide timeout 300 python bench.py --samples $1 --features $2 --steps 10 --warmup 5 --no-cpu-baseline --no-api-leg --verify-rows 100000 > $OUT/bench_mid_$TAG.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/bench_mid_$TAG.json').read().strip().splitlines()[-1])
print('$1 x $2 KMCUDA_AMD_WIDE_MIN_D=$wide', {k:d[k] for k in ('value','ms_per_step')}, d['breakdown_ms_per_step'], d.get('verify',{}).get('ok'))" | tee -a $OUT/mid_widths_$TAG.log
done
done

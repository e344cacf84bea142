#!/bin/bash
# Rows of <= 256 features: the register-resident filters against the LDS-streamed one (KMCUDA_AMD_WIDE_MIN_D=1).
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5t}
for shape in "4000000 256" "1000000 256" "4000000 128" "4000000 64" "4000000 192"; do set -- $shape
for wide in "" 1; do
KMCUDA_AMD_WIDE_MIN_D=$wide timeout 300 python bench.py --samples $1 --features $2 --steps 20 --warmup 5 --no-cpu-baseline --no-api-leg --verify-rows 100000 > $OUT/bench_low_$TAG.json 2>/dev/null
python3 -c "
import json
d=json.loads(open('$OUT/bench_low_$TAG.json').read().strip().splitlines()[-1])
print('$1 x $2 KMCUDA_AMD_WIDE_MIN_D=$wide', {k:d[k] for k in ('value','ms_per_step')}, d['breakdown_ms_per_step'], d.get('verify',{}).get('ok'))" | tee -a $OUT/low_widths_$TAG.log
done
done

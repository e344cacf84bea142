#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_row_cache.py tests/test_gpu_fp16.py tests/test_gpu_lloyd.py -m gpu -q 2>&1 | tail -3
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r4q_$i.json 2>/dev/null; python3 -c "
import json;d=json.loads(open('$OUT/bench_r4q_$i.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['breakdown_ms_per_step'], d.get('verify',{}).get('ok'))"; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1

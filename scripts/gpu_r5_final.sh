#!/bin/bash
# The record of the final build: smoke, the bench line, the files of the suite that the last code change touches.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5final}
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
python3 - <<PY
import json
d=json.loads(open("$OUT/bench_$TAG.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["frac"], d["roofline"]["kernel_ms"], d["breakdown_ms_per_step"], d.get("verify",{}).get("ok"), d["api_kmeans_cuda"]["ms_per_iteration"])
PY
timeout 600 python -m pytest tests/test_gpu_kmeans.py tests/test_gpu_lloyd.py tests/test_gpu_sharded.py -m gpu -q 2>&1 | tail -3

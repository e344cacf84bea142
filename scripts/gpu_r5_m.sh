#!/bin/bash
# Why the second kmeans_cuda() call of a process takes 0.39 s in its loop where the first and third take 0.20.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r5m}
for rep in 1 2; do
KMCUDA_AMD_UPDATE_TRACE=1 timeout 600 python bench.py --api --steps 40 > $OUT/bench_api_trace_$TAG.json 2> $OUT/bench_api_trace_$TAG.err
python3 -c "
import json
d=json.load(open('$OUT/bench_api_trace_$TAG.json'))
print([(c['iterations'],round(c['loop_s'],4),round(c['wall_s'],3),round(c['setup_s'],3)) for c in d['calls']])"
grep update $OUT/bench_api_trace_$TAG.err
done

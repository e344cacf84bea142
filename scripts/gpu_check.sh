#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, the bench line, and the rocprofv3 kernel trace.
# Outputs land in gpurun_out/ (merged back); summaries worth keeping are copied to profiles/ by hand.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
TAG=${1:-r1}
echo "== pytest -m gpu" 
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -5 $OUT/pytest_gpu_$TAG.log
echo "== bench"
timeout 600 python bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; echo "bench rc=$?"
cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
echo "== rocprofv3 kernel trace"
rm -rf $OUT/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o trace -- python bench.py --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
find $OUT/prof_$TAG -name '*stats*' | head

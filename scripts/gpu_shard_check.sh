#!/bin/bash
# Runs on the GPU box (via gpurun): GPU tests, then the N > 1 path of bench.py on ONE GPU (both
# ranks on device 0, gloo instead of RCCL -- the test hooks of bench.py) and one 8-GPU-sized shard.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
TAG=${1:-r1}
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_$TAG.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu_$TAG.log
tail -5 $OUT/pytest_gpu_$TAG.log
echo "== bench --gpus 2, both ranks on GPU 0, gloo"
KMCUDA_AMD_BENCH_SINGLE_DEVICE=1 KMCUDA_AMD_BENCH_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 \
  --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 3 \
  --samples 2000000 > $OUT/bench2_$TAG.json 2> $OUT/bench2_$TAG.err; echo "bench2 rc=$?"
cat $OUT/bench2_$TAG.json; tail -3 $OUT/bench2_$TAG.err
echo "== one 1M-row shard"
timeout 300 python bench.py --samples 1000000 --steps 20 --warmup 10 --no-cpu-baseline > $OUT/bench1m_$TAG.json 2> $OUT/bench1m_$TAG.err; echo "bench1m rc=$?"
cat $OUT/bench1m_$TAG.json; tail -3 $OUT/bench1m_$TAG.err

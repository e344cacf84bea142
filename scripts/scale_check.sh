#!/bin/bash
# scripts/scale_check.sh N [--one-gpu]  -- everything a first run on N GPUs must show, in one command, failing loudly:
#   1. bench.py --gpus N            one process per GPU over torch.distributed (RCCL): the line's n_gpus and
#                                   config.ranks_seen_by_communicator must be N, collectives timed, verify.ok
#   2. bench.py --api --gpus N      whole kmeans_cuda() calls, ONE process, device mask of N GPUs (ncclCommInitAll inside
#                                   the library): shards and ranks must be N
#   3. scripts/scale_parity.py N    kmeans_cuda / knn_cuda over the mask against device 1 (bit-identical assignments,
#                                   k-means++ seeds, neighbour lists; centroids within the fp64 update's rtol 2e-5)
# --one-gpu: no N GPUs here -- both ranks / all shards on GPU 0 (bench: KMCUDA_AMD_BENCH_SINGLE_DEVICE=1 over gloo; the
# library: KMCUDA_AMD_VIRTUAL_SHARDS=N), which runs every line of the sharded host code except RCCL with > 1 rank.
# Output: gpurun_out/scale_<N>[_onegpu]_*.json + one summary line per leg; exit code != 0 if any leg failed.
set -u
cd "$(dirname "$0")/.."
N=${1:?usage: scale_check.sh N [--one-gpu]}
ONE=0; [ "${2:-}" = "--one-gpu" ] && ONE=1
OUT=gpurun_out; mkdir -p $OUT
TAG=scale_${N}; [ $ONE = 1 ] && TAG=${TAG}_onegpu
ROWS=${SCALE_ROWS:-8000000}; STEPS=${SCALE_STEPS:-20}
[ $ONE = 1 ] && ROWS=${SCALE_ROWS:-2000000}
export HSA_ENABLE_IPC_MODE_LEGACY=0
fail=0
check() {   # file, expected ranks, what
python3 - "$1" "$2" "$3" <<'PY'
import json, sys
path, want, what = sys.argv[1], int(sys.argv[2]), sys.argv[3]
try:
    d = json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])
except Exception as e:
    print("FAIL %s: no bench line in %s (%s)" % (what, path, e)); sys.exit(1)
c = d["config"]
seen = c.get("ranks_seen_by_communicator")
shards = c.get("parallelism", "").split("/")[-1]
ok = d["n_gpus"] == want and str(shards) == str(want)
# (--one-gpu runs the library leg on virtual shards: no communicator there, by design)
if "virtual" not in what:
    ok = ok and seen == want
if want > 1 and c.get("collective_ms_per_step") is None:
    ok = False
v = d.get("verify")
if v is not None and not v.get("ok"):
    ok = False
print("%s %s: n_gpus %s, shards %s, ranks_seen_by_communicator %s, %.3f ms per step, %.3e %s, collective %s ms per step%s" % (
    "ok  " if ok else "FAIL", what, d["n_gpus"], shards, seen, d["ms_per_step"], d["value"], d["unit"],
    c.get("collective_ms_per_step"), "" if v is None else ", verify %s" % v.get("ok")))
sys.exit(0 if ok else 1)
PY
}
# 1. one process per GPU
if [ $ONE = 1 ]; then envs="KMCUDA_AMD_BENCH_SINGLE_DEVICE=1 KMCUDA_AMD_BENCH_BACKEND=gloo"; else envs=""; fi
env $envs timeout 1800 python bench.py --gpus $N --samples $ROWS --steps $STEPS --warmup 5 --no-cpu-baseline --verify-rows 200000 \
  > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { echo "FAIL bench.py --gpus $N: rc $? (tail of $OUT/${TAG}_bench.err:)"; tail -5 $OUT/${TAG}_bench.err; fail=1; }
check $OUT/${TAG}_bench.json $N "bench.py --gpus $N (one process per GPU)" || fail=1
# 2. the library's own sharding
if [ $ONE = 1 ]; then envs="KMCUDA_AMD_VIRTUAL_SHARDS=$N"; what="bench.py --api (virtual shards)"; else envs=""; what="bench.py --api --gpus $N (device mask)"; fi
env $envs timeout 1800 python bench.py --api --gpus $N --samples $ROWS --steps $STEPS \
  > $OUT/${TAG}_api.json 2> $OUT/${TAG}_api.err || { echo "FAIL bench.py --api --gpus $N: rc $?"; tail -5 $OUT/${TAG}_api.err; fail=1; }
check $OUT/${TAG}_api.json $N "$what" || fail=1
# 3. parity of whole calls over the mask
if [ $ONE = 1 ]; then flag="--one-gpu"; else flag=""; fi
timeout 1800 python scripts/scale_parity.py $N $flag 2>&1 | tee $OUT/${TAG}_parity.log | tail -6
[ ${PIPESTATUS[0]} = 0 ] || fail=1
[ $fail = 0 ] && echo "scale_check $N: all legs ok" || echo "scale_check $N: FAILED"
exit $fail

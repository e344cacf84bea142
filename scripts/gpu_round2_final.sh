#!/bin/bash
# Round-2 closing session: the whole GPU suite + smoke, then scripts/gpu_round2_d.sh's bench evidence
# (bench line, rocprofv3 kernel stats, 1M-row shard, --api, self-launched 2 ranks over gloo, PMC passes) and
# the BASELINE configs B and D as named.
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r2k}
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_full_$TAG.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_full_$TAG.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
sed -i 's/^timeout 1200 python -m pytest.*$/echo "(suite ran above)"/' scripts/gpu_round2_d.sh
bash scripts/gpu_round2_d.sh $TAG
echo "== config B (Yinyang, Lloyd)"
timeout 300 python scripts/config_b.py --yinyang 0.1 --verbosity 0 2>&1 | grep -o "kmeans_cuda wall.*"
timeout 300 python scripts/config_b.py --yinyang 0 --verbosity 0 2>&1 | grep -o "kmeans_cuda wall.*"
echo "== config D share"
timeout 300 python scripts/config_d.py --samples 8000000 --shard 0/8 --check 100 2>&1 | grep -E "knn_cuda|brute"

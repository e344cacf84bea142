#!/bin/bash
# Yinyang parity tests + config B under rocprofv3 (kernel stats) -> gpurun_out/<tag>_configB_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
TAG=${1:-yy}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_yinyang.py -x -q 2>&1 | tail -3
rm -rf /tmp/yb; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/yb -o b -- python scripts/config_b.py --yinyang 0.1 > /tmp/yb.log 2>&1
grep "wall" /tmp/yb.log
f=$(find /tmp/yb -name "*kernel_stats.csv" | head -1); head -8 $f | cut -c1-150; cp $f gpurun_out/${TAG}_configB_kernel_stats.csv
timeout 300 python scripts/config_b.py --yinyang 0.1 2>&1 | grep wall

#!/bin/bash
# Builds libKMCUDA with -DKMX_YYI_DBG (per-phase s_memtime counters in yy_init_lds_kernel) into scratch/libs/ --
# run this part where hipcc is (no GPU needed) -- and, on the GPU box, runs BASELINE config B with it:
#   bash scripts/yy_init_counters.sh build;  gpurun -- 'bash scripts/yy_init_counters.sh run'
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-build}" = build ]; then
  cd "$ROOT/kmcuda_amd/csrc" && make -s && mkdir -p "$ROOT/scratch/libs"
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-result -DKMX_YYI_DBG -I../../include \
        -c yinyang_init.hip -o /tmp/yyi_dbg.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/scratch/libs/libyyidbg.so" $(ls *.o | grep -v '^yinyang_init.o$') /tmp/yyi_dbg.o
else
  cd "$ROOT" && KMCUDA_AMD_LIB="$ROOT/scratch/libs/libyyidbg.so" python scripts/yy_init_counters.py --yinyang 0.1 --verbosity 0
fi

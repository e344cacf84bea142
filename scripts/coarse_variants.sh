#!/bin/bash
# Builds side-by-side variants of libKMCUDA.so that differ only in the stage-1 kernel's experiment
# switches (lloyd_f16.hip: KMX_BOOK / KMX_BIASPF / KMX_PRIO / KMX_TRACE / KMX_ABL / KMX_PD) into
# scratch/libs/lib<name>.so, and prints the kernel's register / scratch usage.  Runs here (no GPU).
#   scripts/coarse_variants.sh name1:"-DKMX_BOOK=1" name2:"-DKMX_BOOK=1 -DKMX_BIASPF=1" ...
set -e
cd "$(dirname "$0")/../kmcuda_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../scratch/libs
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include $flags \
    -Rpass-analysis=kernel-resource-usage -c lloyd_f16.hip -o /tmp/lloyd_f16_$name.o 2> /tmp/lloyd_f16_$name.log
  objs=$(ls *.o | grep -v '^lloyd_f16.o$' | tr '\n' ' ')
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scratch/libs/lib$name.so $objs /tmp/lloyd_f16_$name.o
  echo "== $name ($flags)"
  grep -A12 "lloyd_coarse2_kernelILi256ELb0ELb1ELb1ELi2E" /tmp/lloyd_f16_$name.log | grep -E "VGPRs:|AGPRs|Scratch|Occupancy|LDS" | sed 's/.*remark: [^ ]* //' | tr '\n' ' '
  echo
done

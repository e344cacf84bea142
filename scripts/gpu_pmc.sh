#!/bin/bash
# PMC passes for the bench's dominant kernel (separate runs per counter group, kernel-trace only).
set -u
export TMPDIR=/tmp
OUT=gpurun_out
TAG=${1:-r1}
mkdir -p $OUT
CMD="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf $OUT/pmc_${TAG}_$name
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_${TAG}_$name -o pmc -- $CMD > $OUT/pmc_${TAG}_$name.log 2>&1
  echo "pmc $grp rc=$?"
done
find $OUT -name '*counter_collection.csv' | head

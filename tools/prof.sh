#!/bin/bash
# Usage (on the GPU box, from the repo root): bash tools/prof.sh <tag> -- <bench args>
# Writes rocprofv3 kernel-trace stats and separate PMC passes under gpurun_out/<tag>/.
set -u
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python $R/bench.py "$@" --no-cpu-baseline --no-single-bag > $OUT/trace.log 2>&1
for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout 600 rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/pmc_$n -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-single-bag > $OUT/pmc_$n.log 2>&1
done
find $OUT -name "*.csv" | head -40

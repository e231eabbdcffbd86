#!/bin/bash
# GPU-box driver script of round 6 (run through gpurun from the repo root): TAG=<dir> bash tools/r6.sh <stage> ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG:-r6}
mkdir -p $OUT
cd $R
prof() {  # prof <name> <cmd...>: rocprofv3 kernel stats of a command -> $OUT/<name>_stats.csv
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_$name -o t -- "$@" > $OUT/trace_$name.log 2>&1)
  f=$(find $OUT/trace_$name -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${name}_stats.csv && head -${PROF_LINES:-30} $OUT/${name}_stats.csv | cut -c1-160
}
for stage in "$@"; do
case $stage in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log;;
tests_agg)
  timeout 900 python -m pytest tests/test_agg_bwd_gpu.py tests/test_agg_gpu.py tests/test_agg_bf16_gpu.py -m gpu -x -q > $OUT/pytest_agg.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_agg.log;;
tests_bwd)
  timeout 900 python -m pytest tests/test_agg_bwd_gpu.py -m gpu -x -q > $OUT/pytest_bwd.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_bwd.log;;
tests_emb)
  timeout 900 python -m pytest tests/test_resnet_gpu.py tests/test_forms_gpu.py -m gpu -x -q > $OUT/pytest_emb.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_emb.log;;
bench)
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err;;
bench_train)
  timeout 300 python bench.py --workload train --no-cpu-baseline > $OUT/bench_train.json 2> $OUT/bench_train.err; echo "bench rc=$?"; tail -c 1200 $OUT/bench_train.json; tail -3 $OUT/bench_train.err;;
bench_emb)
  timeout 300 python bench.py --workload embedder --no-cpu-baseline > $OUT/bench_emb.json 2> $OUT/bench_emb.err; echo "bench rc=$?"; tail -c 900 $OUT/bench_emb.json; tail -3 $OUT/bench_emb.err;;
bench_agg)
  timeout 300 python bench.py --workload aggregator,aggregator_bf16 --no-cpu-baseline > $OUT/bench_agg.json 2> $OUT/bench_agg.err; echo "bench rc=$?"; tail -c 900 $OUT/bench_agg.json; tail -3 $OUT/bench_agg.err;;
prof_train)
  prof train python $R/tools/train_bench.py --classes ${CLASSES:-1} --steps 40;;
prof_fused)
  prof fused python $R/tools/train_fused.py ${CLASSES:-1} 200;;
prof_single)
  prof single python $R/tools/single_bag.py c16;;
prof_agg)
  prof agg python $R/bench.py --workload aggregator,aggregator_bf16 --streams 1 --steps 4 --warmup 1 --min-seconds 0.1 --no-cpu-baseline --no-single-bag;;
prof_emb)
  prof emb python $R/bench.py --workload embedder --streams 1 --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline;;
pmc_fused)
  for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-40)
    (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/pmc_fused_$n -o p -- python $R/tools/train_fused.py 1 60 > $OUT/pmc_fused_$n.log 2>&1)
  done; find $OUT -name "*counter_collection.csv" | head;;
pmc_single)
  for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-40)
    (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/pmc_single_$n -o p -- python $R/tools/single_bag.py c16 > $OUT/pmc_single_$n.log 2>&1)
  done; find $OUT -name "*counter_collection.csv" | head;;
prof_emb16)
  prof emb16 python $R/bench.py --workload embedder_half --streams 1 --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline;;
pmc_agg|pmc_emb|pmc_emb16)
  W=aggregator,aggregator_bf16; [ $stage = pmc_emb ] && W=embedder; [ $stage = pmc_emb16 ] && W=embedder_half
  for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-40)
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/${stage}_$n -o p -- python $R/bench.py --workload $W --streams 1 --steps 3 --warmup 1 --min-seconds 0.05 --no-cpu-baseline --no-single-bag > $OUT/${stage}_$n.log 2>&1)
  done; find $OUT -name "*counter_collection.csv" | head;;
*) echo "unknown stage $stage";;
esac
done

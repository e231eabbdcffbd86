#!/bin/bash
# GPU-box driver script of round 3 (run through gpurun from the repo root): bash tools/r3.sh <stage> ...
# stages: tests | tests_emb | tests_bf16 | tests_new | bench | variants | variants2 | variants_pp | variants_alt | variants_wide |
#         stamps | stamps_wino | prof_agg | prof_emb | pmc_agg | pmc_emb    (variants* and stamps* need the expt / trace builds:
#         python dsmil-wsi_amd/build.py --variant expt -DDSMIL_EXPERIMENTS ; --variant trace -DDSMIL_EXPERIMENTS -DDSMIL_TRACE)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${TAG:-r3}
mkdir -p $OUT
cd $R
for stage in "$@"; do
case $stage in
tests)
  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log; tail -5 $OUT/pytest_gpu.log;;
bench)
  timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err;;
variants)
  DSMIL_NATIVE_LIB=libdsmil_hip_expt.so timeout 900 python tools/variants.py aggregator base: xe:DSMIL_EXPT=8 mlponly:DSMIL_EXPT=4 xe_mlponly:DSMIL_EXPT=12 novsum:DSMIL_EXPT=1 s9:DSMIL_MLP=s9 > $OUT/variants_agg.log 2>&1; cat $OUT/variants_agg.log;;
variants2)
  DSMIL_NATIVE_LIB=libdsmil_hip_expt.so timeout 900 python tools/variants.py aggregator base: tu16:DSMIL_EXPT=16 tu32:DSMIL_EXPT=32 > $OUT/variants_agg2.log 2>&1; cat $OUT/variants_agg2.log
  DSMIL_NATIVE_LIB=libdsmil_hip_expt.so VARIANT_ROUNDS=1 timeout 900 python tools/variants.py embedder base: noxform:DSMIL_WINO_EXPT=1 noraw:DSMIL_WINO_EXPT=2 nou:DSMIL_WINO_EXPT=4 noepi:DSMIL_WINO_EXPT=8 nomfma:DSMIL_WINO_EXPT=16 onlymfma:DSMIL_WINO_EXPT=15 > $OUT/variants_emb.log 2>&1; cat $OUT/variants_emb.log;;
tests_bf16)
  timeout 900 python -m pytest tests/test_agg_bf16_gpu.py tests/test_resnet_gpu.py tests/test_forms_gpu.py -m gpu -x -q > $OUT/pytest_bf16.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_bf16.log;;
tests_emb)
  timeout 900 env DSMIL_WINO_KERNEL=${WINO_KERNEL:-unit} python -m pytest tests/test_resnet_gpu.py tests/test_forms_gpu.py -m gpu -x -q > $OUT/pytest_emb.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_emb.log;;
variants_pp)
  DSMIL_NATIVE_LIB=libdsmil_hip_expt.so VARIANT_ROUNDS=${VARIANT_ROUNDS:-2} timeout 900 python tools/variants.py embedder unit: pp:DSMIL_WINO_KERNEL=pp alt:DSMIL_WINO_KERNEL=alt > $OUT/variants_pp.log 2>&1; cat $OUT/variants_pp.log;;
stamps_wino)
  for k in ${STAMP_K:-1 12}; do DSMIL_WINO_KERNEL=${WINO_KERNEL:-unit} DSMIL_NATIVE_LIB=libdsmil_hip_trace.so DSMIL_WINO_TRACE=$k timeout 300 python tools/stamp_wino.py > $OUT/stamps_wino_$k.log 2>&1; echo "== launch $k"; tail -14 $OUT/stamps_wino_$k.log | cut -c1-400; done;;
variants_wide)
  DSMIL_NATIVE_LIB=libdsmil_hip_expt.so VARIANT_ROUNDS=${VARIANT_ROUNDS:-2} timeout 900 python tools/variants.py embedder wide: narrow:DSMIL_WINO_NARROW=1 > $OUT/variants_wide.log 2>&1; cat $OUT/variants_wide.log;;
variants_alt)
  DSMIL_NATIVE_LIB=libdsmil_hip_expt.so VARIANT_ROUNDS=${VARIANT_ROUNDS:-2} timeout 900 python tools/variants.py embedder wide: alt:DSMIL_WINO_KERNEL=alt > $OUT/variants_alt.log 2>&1; cat $OUT/variants_alt.log;;
tests_new)
  timeout 900 python -m pytest tests/test_agg_bwd_gpu.py tests/test_agg_gpu.py tests/test_entry_points.py -m gpu -x -q > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_new.log;;
stamps)
  for e in 68 76; do DSMIL_NATIVE_LIB=libdsmil_hip_trace.so DSMIL_EXPT=$e timeout 300 python tools/stamp.py > $OUT/stamps_$e.log 2>&1; tail -12 $OUT/stamps_$e.log; done;;
prof_agg)
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_agg -o t -- python $R/bench.py --workload aggregator,aggregator_bf16 --streams 1 --steps 4 --warmup 1 --min-seconds 0.1 --no-cpu-baseline --no-single-bag > $OUT/trace_agg.log 2>&1); find $OUT/trace_agg -name "*kernel_stats.csv" | head -2 | xargs -I{} sh -c 'head -12 {}';;
prof_emb)
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace_emb -o t -- python $R/bench.py --workload embedder --streams 1 --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline > $OUT/trace_emb.log 2>&1); find $OUT/trace_emb -name "*kernel_stats.csv" | head -2 | xargs -I{} sh -c 'head -24 {}';;
pmc_agg|pmc_emb)
  W=aggregator,aggregator_bf16; [ $stage = pmc_emb ] && W=embedder
  for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
    n=$(echo $c | tr ' ' '_' | cut -c1-40)
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/${stage}_$n -o p -- python $R/bench.py --workload $W --streams 1 --steps 3 --warmup 1 --min-seconds 0.05 --no-cpu-baseline --no-single-bag > $OUT/${stage}_$n.log 2>&1)
  done; find $OUT -name "*counter_collection.csv" | head;;
esac
done

cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; tail -c 400 gpurun_out/bench_full.json
for m in f32 s6; do DSMIL_MLP=$m timeout 300 python bench.py --workload aggregator --no-cpu-baseline 2>/dev/null | grep "^{" > gpurun_out/bench_agg_$m.json; done
for m in f32 s9 s6; do DSMIL_MLP=$m timeout 300 python tests/accuracy_report.py 2>&1 | grep mode; done > gpurun_out/accuracy_mlp_forms.jsonl
bash tools_prof.sh agg_v3 -- --workload aggregator --steps 10 --warmup 2 > /dev/null 2>&1
python tools_prof_summary.py gpurun_out/agg_v3 gpurun_out/agg_v3_sum 2>&1 | tail -6
timeout 300 python tools_train_bench.py --steps 100 2>&1 | tail -1 > gpurun_out/train_bench.json; cat gpurun_out/train_bench.json

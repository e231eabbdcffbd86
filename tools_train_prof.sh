R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/train
cd $R && timeout 300 python tools_train_bench.py --steps 100 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/train/trace -o t -- python $R/tools_train_bench.py --steps 50 > $R/gpurun_out/train/trace.log 2>&1
head -40 $R/gpurun_out/train/trace/*/t_kernel_stats.csv 2>/dev/null || find $R/gpurun_out/train -name "*stats*"
cd $R && timeout 200 python bench.py --workload aggregator --steps 50 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900

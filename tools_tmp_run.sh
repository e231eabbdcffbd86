cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -i "smoke\|error\|assert" | head
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; python -c "
import json; d=json.load(open('gpurun_out/bench_full.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['whole_path_frac_of_roofline'], d['config']['single_bag_forward_ms'], d['embedder']['value'], d['cpu_baseline']['value'], d['embedder']['cpu_baseline']['value'])"

cd ${GRAFT_REPO_ROOT:-.}
for m in s9 s6; do DSMIL_MLP=$m timeout 300 python bench.py --workload aggregator --no-cpu-baseline --steps 40 2>/dev/null | grep "^{" | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$m', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"; done
timeout 900 python -m pytest tests/test_agg_gpu.py -m gpu -q 2>&1 | tail -2

cd ${GRAFT_REPO_ROOT:-.}
DSMIL_WINO_DEBUG=1 timeout 200 python bench.py --workload embedder --no-cpu-baseline --steps 3 --warmup 1 2>&1 | grep "dsmil\]" | head -3

#!/bin/bash
# Usage (GPU box): bash tools_trace.sh <tag> -- <bench args>; kernel-trace only, prints per-kernel averages
set -u
TAG=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o t -- python $R/bench.py "$@" --no-cpu-baseline > $OUT/trace.log 2>&1
tail -2 $OUT/trace.log | head -1 | cut -c1-400
python - <<PY
import csv,collections
rows=list(csv.DictReader(open("$OUT/trace/t_kernel_trace.csv")))
agg=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if 'at::' in n or 'rocclr' in n: continue
    nm=n.split('(')[0].replace('void ','').replace('(anonymous namespace)::','')
    key=(nm, r['Grid_Size_X'], r['Grid_Size_Y'])
    agg.setdefault(key,[]).append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in agg.items():
    print(f"{k[0]:34s} grid=({k[1]:>8s},{k[2]:>4s}) n={len(v):4d} avg={sum(v)/len(v)/1e3:9.1f} us total={sum(v)/1e6:8.2f} ms")
PY

#!/bin/bash
# per-layer durations of the direct convs (k_conv_s6) under forced tile shapes: DSMIL_S6_TILE = 0 (heuristic) 44 42 24 22
# (experiment build).  bash tools/s6_shapes.sh   -> gpurun_out/s6/<shape>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/s6
mkdir -p $OUT
for t in 0 44 42 24 22; do
  (cd /tmp && export TMPDIR=/tmp && DSMIL_NATIVE_LIB=libdsmil_hip_expt.so DSMIL_S6_TILE=$t timeout 300 rocprofv3 --kernel-trace -f csv -d $OUT/t$t -o t -- python $R/bench.py --workload embedder --streams 1 --steps 2 --warmup 1 --min-seconds 0.05 --no-cpu-baseline --no-single-bag > $OUT/t$t.log 2>&1)
  python - $OUT/t$t/t_kernel_trace.csv $t <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
seq = [(r['Kernel_Name'][r['Kernel_Name'].find('k_conv_s6'):r['Kernel_Name'].find('(')], int(r['Grid_Size_X']) // 256, int(r['Grid_Size_Y']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'k_conv_s6' in r['Kernel_Name']]
# six direct convs per forward, in launch order
n = len(seq) // 6
per = collections.defaultdict(list)
for i, s in enumerate(seq):
    per[i % 6].append(s)
tot = 0.0
for k in range(6):
    v = sorted(x[3] for x in per[k]); med = v[len(v) // 2]; tot += med
    print(sys.argv[2], 'conv', k, per[k][0][:3], 'median %.1f us' % med)
print(sys.argv[2], 'sum %.1f us' % tot)
P
done

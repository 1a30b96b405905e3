#!/bin/bash
# Which kernels surround the tiny torch fill kernels of an eager training step (rocprofv3 kernel trace, ordered by start).
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -o kt -- python $R/bench.py --steps 1 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $R/gpurun_out/fill_context.log 2>&1
F=$(find /tmp/kt -name "*kernel_trace.csv" | head -1)
python - "$F" >> $R/gpurun_out/fill_context.log 2>&1 <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n[:48]
ctx = collections.Counter()
sizes = collections.Counter()
for i, r in enumerate(rows):
    if 'FillFunctor<float>' in r['Kernel_Name']:
        prev = short(rows[i - 1]['Kernel_Name']) if i else ''
        nxt = short(rows[i + 1]['Kernel_Name']) if i + 1 < len(rows) else ''
        ctx[(prev, nxt)] += 1
        sizes[(r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', '?'))] += 1
print('fill kernels:', sum(ctx.values()), 'of', len(rows))
for k, v in ctx.most_common(30):
    print(v, k)
print('grid sizes:', sizes.most_common(12))
PY
tail -45 $R/gpurun_out/fill_context.log

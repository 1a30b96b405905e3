#!/bin/bash
# rocprofv3 kernel trace of tools/chain_bench.py on a few shapes; prints the per-kernel durations of the last graph replays
cd "$(dirname "$0")/.." || exit 1
R=$(pwd); OUT=gpurun_out/${1:-pc}; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/pc
timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/pc -o t -- python $R/tools/chain_bench.py --reps 3 --shapes ${2:-316x64,64x256} > $R/$OUT/prof.log 2>&1
cp /tmp/pc/t_kernel_trace.csv $R/$OUT/ 2>/dev/null
cd $R
python - <<'PY' $OUT/t_kernel_trace.csv
import csv,sys,re,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
agg=collections.OrderedDict()
for r in rows:
    n=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name']); n=re.sub(r'^void ','',n)[:60]
    k=(n, r['Grid_Size_X'], r['Grid_Size_Y'])
    a=agg.setdefault(k,[0,0.0,1e9]); d=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    a[0]+=1; a[1]+=d; a[2]=min(a[2],d)
for k,v in agg.items():
    if 'chain' in k[0] or 'flash' in k[0]:
        print('%4d x avg %8.1f min %8.1f us  grid %s x %s  %s'%(v[0],v[1]/v[0],v[2],k[1],k[2],k[0]))
PY

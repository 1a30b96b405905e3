#!/bin/bash
# Second PMC pass over tools/gemm_pmc_driver.py: where the issue cycles of the split GEMM go (VALU / LDS / VMEM instruction counts and
# active cycles, LDS bank conflicts).  Counters only (no tracing domains).  Usage: [RIH_PMC_ENGINE=2] [RIH_PMC_PRESPLIT=1] tools/gemm_pmc2.sh <outdir>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-gemm_pmc2}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmcg2
CNT=${RIH_PMC_COUNTERS:-SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT}
timeout 600 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmcg2 -o pmc -- python $R/tools/gemm_pmc_driver.py > $OUT/driver.log 2>&1
f=$(find /tmp/pmcg2 -name "*counter_collection*.csv" | head -1)
python - "$f" "$OUT/driver.log" > $OUT/gemm_pmc2_table.txt <<'PY'
import csv, collections, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
plan = [json.loads(l) for l in open(sys.argv[2]) if l.startswith('[{')][0]
disp = collections.OrderedDict()
names = []
for r in rows:
    k = r['Kernel_Name']
    if 'gemm_split_kernel' not in k and 'gemm_split256' not in k:
        continue
    d = disp.setdefault(int(r['Dispatch_Id']), {'dur': int(r['End_Timestamp']) - int(r['Start_Timestamp']), 'waves': int(r['Grid_Size']) // 64})
    d[r['Counter_Name']] = float(r['Counter_Value'])
    if r['Counter_Name'] not in names:
        names.append(r['Counter_Name'])
ds = [disp[k] for k in sorted(disp)]
i = 0
print('per WAVE averages (counter / waves); SQ_WAVE_CYCLES, SQ_ACTIVE_*, SQ_WAIT_* in quad-cycles')
print('%-74s %8s %7s ' % ('variant', 'dur us', 'waves') + ' '.join('%14s' % n[-14:] for n in names))
for p in plan:
    g = ds[i:i + p['count']][2:]
    i += p['count']
    m = lambda c: sum(x.get(c, 0.0) for x in g) / len(g)
    w = g[0]['waves']
    print('%-74s %8.1f %7d ' % (p['label'][:74], m('dur') / 1e3, w) + ' '.join('%14.0f' % (m(n) / w) for n in names))
PY
cat $OUT/gemm_pmc2_table.txt

#!/bin/bash
# PMC pass (cycle counters only, own run) over tools/gemm_pmc_driver.py: per shape clock, MFMA-busy, wait breakdown of the production GEMM.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/${1:-gemm_pmc}
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmcg
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcg -o pmc -- python $R/tools/gemm_pmc_driver.py > $OUT/driver.log 2>&1
f=$(find /tmp/pmcg -name "*counter_collection*.csv" | head -1)
cp "$f" $OUT/gemm_counter_collection.csv
python - "$f" "$OUT/driver.log" > $OUT/gemm_pmc_table.txt <<'PY'
import csv, collections, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
plan = [json.loads(l) for l in open(sys.argv[2]) if l.startswith('[{')][0]      # rocprofv3 prints after the driver's last line
disp = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name']
    if 'gemm_p3_kernel' not in k and 'gemm_split_kernel' not in k and 'gemm_split256' not in k:
        continue
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': k, 'dur': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
    d[r['Counter_Name']] = float(r['Counter_Value'])
ds = [disp[k] for k in sorted(disp)]
i = 0
print('%-78s %9s %8s %9s %8s %8s %8s %8s' % ('variant', 'dur us', 'clk GHz', 'mfma-busy', 'TF', 'wait_any', 'wait_ins', 'active'))
for p in plan:
    g = ds[i:i + p['count']][2:]      # skip two warm-up launches
    i += p['count']
    m = lambda c: sum(x[c] for x in g) / len(g)
    dur = m('dur')
    cyc = m('GRBM_GUI_ACTIVE') / 8
    wc = m('SQ_WAVE_CYCLES')
    print('%-78s %9.1f %8.2f %8.1f%% %8.1f %7.1f%% %7.1f%% %7.1f%%' % (p['label'], dur / 1e3, cyc / dur,
          100 * m('SQ_VALU_MFMA_BUSY_CYCLES') / (cyc * 1024), p['flop'] / dur / 1e3, 100 * m('SQ_WAIT_ANY') / wc,
          100 * m('SQ_WAIT_INST_ANY') / wc, 100 * m('SQ_ACTIVE_INST_ANY') / wc))
PY
cat $OUT/gemm_pmc_table.txt

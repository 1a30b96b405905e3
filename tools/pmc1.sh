#!/bin/bash
# one PMC pass (cycle counters) over a gemm_bench mode; prints per-kernel clock and MFMA utilisation
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmcx
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmcx -o pmc -- python $R/tools/gemm_bench.py ${PMC_MODE:---dbg} > $R/gpurun_out/pmc/runx.log 2>&1
f=$(find /tmp/pmcx -name "*counter_collection*.csv" | head -1)
cp $f $R/gpurun_out/pmc/setx_counter_collection.csv
python - <<PY
import csv,collections
rows=list(csv.DictReader(open("$f")))
agg=collections.OrderedDict()
for r in rows:
    k=r['Kernel_Name']
    if 'gemm' not in k: continue
    key=k[k.index('gemm'):k.index('>')+1]+' g'+r['Grid_Size']
    d=agg.setdefault(key,{'dur':[], })
    d.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    if r['Counter_Name']=='GRBM_GUI_ACTIVE': d['dur'].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,d in agg.items():
    m=lambda c: sum(d[c])/len(d[c])
    cyc=m('GRBM_GUI_ACTIVE')/8; dur=m('dur')
    print('%-50s dur %7.1fus clk %.2fGHz mfma_util %.1f%% wait_any %.2e wait_inst %.2e active %.2e wavecyc %.2e'%(k,dur/1e3,cyc/dur, 100*m('SQ_VALU_MFMA_BUSY_CYCLES')/(cyc*1024), m('SQ_WAIT_ANY'), m('SQ_WAIT_INST_ANY'), m('SQ_ACTIVE_INST_ANY'), m('SQ_WAVE_CYCLES')))
PY

#!/bin/bash
# Whole-training-step PMC evidence (eager launches, so every kernel is a separate dispatch):
#   pass 1: SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE  -> MFMA-busy fraction and clock per kernel family
#   pass 2/3: FETCH_SIZE, WRITE_SIZE                     -> HBM traffic (tools/pmc_traffic.sh does the same)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc_step
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_step -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph > $R/gpurun_out/pmc/step_mfma.log 2>&1
f=$(find /tmp/pmc_step -name "*counter_collection*.csv" | head -1)
python - "$f" <<'PY' > $R/gpurun_out/pmc/step_mfma.txt
import csv, sys, collections
agg = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'gemm' in k:
        name = k[k.index('gemm'):k.index('>') + 1]
    else:
        name = 'non-GEMM kernels'
    d = agg.setdefault(name, collections.defaultdict(float))
    d[r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        d['ns'] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        d['n'] += 1
print('rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over 3 eager training steps (bench.py --no-graph, B=64)')
print('mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 1024 SIMDs); clk = GRBM_GUI_ACTIVE/8 / duration')
tot = collections.defaultdict(float)
rows = []
for name, d in agg.items():
    cyc = d['GRBM_GUI_ACTIVE'] / 8.0
    rows.append((d['ns'], name, d['n'], 100.0 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / max(cyc * 1024, 1), cyc / max(d['ns'], 1)))
    if name.startswith('gemm'):
        for k2 in ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'ns', 'n'):
            tot[k2] += d[k2]
for ns, name, n, busy, clk in sorted(rows, reverse=True):
    print('%9.2f ms  %6d launches  mfma_busy %5.1f%%  clk %.2f GHz  %s' % (ns / 1e6, n, busy, clk, name))
cyc = tot['GRBM_GUI_ACTIVE'] / 8.0
print('ALL rih_gemm kernels: %.2f ms, %d launches, mfma_busy %.1f%%, clk %.2f GHz'
      % (tot['ns'] / 1e6, tot['n'], 100.0 * tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024), cyc / tot['ns']))
PY
head -30 $R/gpurun_out/pmc/step_mfma.txt
bash $R/tools/pmc_traffic.sh > /dev/null 2>&1
head -2 $R/gpurun_out/pmc/traffic_FETCH_SIZE.txt; head -2 $R/gpurun_out/pmc/traffic_WRITE_SIZE.txt

#!/bin/bash
# PMC pass (cycle counters only, own run: never together with a trace domain other than --kernel-trace) over a driver script;
# per (kernel, grid) = per shape: average duration, effective clock, MFMA-busy, wait breakdown, VALU instructions per wavefront.
# usage: tools/r5_pmc.sh <out-name> <kernel-regex> <driver.py> [env assignments ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; PAT=$2; DRV=$3; shift 3
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc5
env "$@" timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc5 -o pmc -- python $R/$DRV > $OUT/driver.log 2>&1
f=$(find /tmp/pmc5 -name "*counter_collection*.csv" | head -1)
python - "$f" "$PAT" > $OUT/pmc_table.txt <<'PY'
import csv, collections, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = re.compile(sys.argv[2])
disp = collections.OrderedDict()
for r in rows:
    k = r['Kernel_Name']
    if not pat.search(k):
        continue
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': re.sub(r'\(anonymous namespace\)::|^void ', '', k)[:64], 'grid': r.get('Grid_Size', '?'),
                                                 'dur': int(r['End_Timestamp']) - int(r['Start_Timestamp'])})
    d[r['Counter_Name']] = float(r['Counter_Value'])
agg = collections.OrderedDict()
for d in disp.values():
    agg.setdefault((d['name'], d['grid']), []).append(d)
print('%-66s %9s %5s %9s %8s %9s %8s %8s %8s %10s' % ('kernel', 'grid', 'n', 'dur us', 'clk GHz', 'mfma-busy', 'wait_any', 'wait_ins', 'active', 'valu/wave'))
for (name, grid), g in agg.items():
    g = g[len(g) // 4:]          # drop the first quarter (warm-up launches)
    m = lambda c: sum(x.get(c, 0.0) for x in g) / len(g)
    dur, cyc, wc = m('dur'), m('GRBM_GUI_ACTIVE') / 8, max(m('SQ_WAVE_CYCLES'), 1.0)
    waves = int(grid) / 64.0 if str(grid).isdigit() else 1.0
    print('%-66s %9s %5d %9.1f %8.2f %8.1f%% %7.1f%% %7.1f%% %7.1f%% %10.0f' % (name, grid, len(g), dur / 1e3, cyc / max(dur, 1), 100 * m('SQ_VALU_MFMA_BUSY_CYCLES') / max(cyc * 1024, 1),
          100 * m('SQ_WAIT_ANY') / wc, 100 * m('SQ_WAIT_INST_ANY') / wc, 100 * m('SQ_ACTIVE_INST_ANY') / wc, m('SQ_INSTS_VALU') / max(waves, 1)))
PY
cat $OUT/pmc_table.txt | cut -c1-200

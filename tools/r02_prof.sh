#!/bin/bash
# rocprofv3 kernel summary of the headline step (own run: kernel trace + stats only)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/${1:-r02_prof}
mkdir -p "$OUT"
cd /tmp
rm -rf /tmp/prof_step
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-loop > $OLDPWD/$OUT/prof_bench.log 2>&1
cd $OLDPWD
find /tmp/prof_step -name '*kernel_stats*.csv' -exec cp {} "$OUT/bench_kernel_stats.csv" \;
ls /tmp/prof_step/* | head
tail -2 "$OUT/prof_bench.log" | cut -c1-300
python - "$OUT/bench_kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.1f ms over the profiled run' % (tot / 1e6))
for r in rows[:40]:
    print('%-90s calls %6s total %8.2f ms %5.1f%% avg %8.1f us' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e6, 100 * float(r['TotalDurationNs']) / tot, float(r['AverageNs']) / 1e3))
PY

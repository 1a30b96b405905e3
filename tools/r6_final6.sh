#!/bin/bash
# Round 6, last evidence session (ABI 19 head): whole GPU suite, smoke, default bench line, HRNet-W32 line and its kernel trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd)
O=gpurun_out/r6final6; mkdir -p $O
OUT=r6final6 bash tools/r6_final3.sh
( timeout 600 python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop ) > $O/bench_hrnet.log 2>&1; grep '^{' $O/bench_hrnet.log | cut -c1-200
cd /tmp; rm -rf /tmp/trh
( time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trh -o t -- python $R/bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline ) > $R/$O/prof_bench_hrnet.log 2>&1
cd $R
T=$(find /tmp/trh -name "*kernel_trace.csv" | head -1); S=$(find /tmp/trh -name "*kernel_stats.csv" | head -1)
cp "$S" $O/hrnet_kernel_stats.csv 2>/dev/null
python tools/step_from_trace.py "$T" --top 50 --by-grid > $O/step_trace_hrnet.txt 2>&1
head -8 $O/step_trace_hrnet.txt | cut -c1-170
echo done

#!/bin/bash
# Round 6: kernel trace of the default bench command (per kernel and per grid), optional env via PROF_ENV, output tag via OUT.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); O=gpurun_out/${OUT:-r6prof}; mkdir -p $O
cd /tmp; rm -rf /tmp/tr
( time timeout 600 env $PROF_ENV rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline $BENCH_ARGS ) > $R/$O/prof_bench.log 2>&1
cd $R
T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); S=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
cp "$S" $O/bench_kernel_stats.csv 2>/dev/null
python tools/step_from_trace.py "$T" --top 80 --by-grid > $O/step_trace.txt 2>&1
head -60 $O/step_trace.txt | cut -c1-170
grep '^{' $O/prof_bench.log | tail -1 | cut -c1-160

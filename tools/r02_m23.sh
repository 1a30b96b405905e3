#!/bin/bash
# kernel trace of the headline step (rocprofv3 kernel trace + stats only); the full trace is kept so that ONE graph replay can be
# cut out of it (tools/step_from_trace.py)
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r02_m23
mkdir -p "$OUT"
cd /tmp
rm -rf /tmp/prof_step
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-loop ) > $R/$OUT/prof_bench.log 2>&1
cp /tmp/prof_step/step_kernel_stats.csv $R/$OUT/bench_kernel_stats.csv
cp /tmp/prof_step/step_kernel_trace.csv $R/$OUT/bench_kernel_trace.csv
ls -la /tmp/prof_step
tail -1 $R/$OUT/prof_bench.log | cut -c1-200

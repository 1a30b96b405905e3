#!/bin/bash
# Round 5, GPU session 20: kernel trace of the HRNet-W32 step (family breakdown: BatchNorm, GEMM, finishers).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd); O=gpurun_out/r5c20; mkdir -p $O
cd /tmp; rm -rf /tmp/trh
( time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trh -o t -- python $R/bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline ) > $R/$O/prof_bench_hrnet.log 2>&1
cd $R
T=$(find /tmp/trh -name "*kernel_trace.csv" | head -1); S=$(find /tmp/trh -name "*kernel_stats.csv" | head -1)
cp "$S" $O/hrnet_kernel_stats.csv 2>/dev/null
python tools/step_from_trace.py "$T" --top 70 --mark nchw_to_nhwc --by-grid > $O/step_trace_hrnet.txt 2>&1
head -50 $O/step_trace_hrnet.txt | cut -c1-150
grep '^{' $O/prof_bench_hrnet.log | tail -1 | cut -c1-160

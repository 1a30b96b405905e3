#!/bin/bash
# Round 6, after the rih_gemm prune (kernel names lost two template arguments): the default bench line and the kernel trace of the
# same command on the final library, so that profiles/ names the kernels the tree builds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd)
O=gpurun_out/${OUT:-r6trace}; mkdir -p $O
( timeout 600 python bench.py ) > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log | cut -c1-200
cd /tmp; rm -rf /tmp/tr
( time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/bench.py --no-cpu-baseline ) > $R/$O/prof_bench_default.log 2>&1
cd $R
T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); S=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
cp "$S" $O/bench_kernel_stats.csv 2>/dev/null
python tools/step_from_trace.py "$T" --top 70 --by-grid > $O/step_trace.txt 2>&1
head -4 $O/step_trace.txt | cut -c1-160
( timeout 600 python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop ) > $O/bench_hrnet.log 2>&1; grep '^{' $O/bench_hrnet.log | cut -c1-200
echo done

#!/bin/bash
# Round 5, GPU session 7: kernel trace of the training bench, GEMM-family launches of one replay by (kernel, grid) = by shape.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c7; mkdir -p $O
R=$(pwd)
cd /tmp; rm -rf /tmp/tr
( time timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $R/bench.py --no-cpu-baseline --no-reference-loop --no-roofline --steps 10 --warmup 3 ) > $R/$O/prof_bench.log 2>&1
cd $R
T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python tools/step_from_trace.py "$T" --top 5 --by-grid > $O/step_by_grid.txt 2>&1
sed -n '/GEMM-family launches by/,$p' $O/step_by_grid.txt | cut -c1-190
echo done

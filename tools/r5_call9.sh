#!/bin/bash
# Round 5, GPU session 9: PMC tables of the new kernels (MFMA-busy, clock, waits), HBM traffic of the GEMM family by PMC (traffic_v6),
# the default bench line, rocprofv3 kernel trace + stats of the default command.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd)
O=gpurun_out/r5c9; mkdir -p $O
bash tools/r5_pmc.sh r5c9/pmc_conv3 "conv3x3_halo_kernel|gemm_split_kernel" tools/conv3_bench.py > $O/pmc_conv3.out 2>&1
cut -c1-200 gpurun_out/r5c9/pmc_conv3/pmc_table.txt
bash tools/r5_pmc.sh r5c9/pmc_panel "panel_kernel|gemm_split_kernel" tools/panel_bench.py > $O/pmc_panel.out 2>&1
cut -c1-200 gpurun_out/r5c9/pmc_panel/pmc_table.txt
bash tools/pmc_traffic.sh > $O/pmc_traffic.out 2>&1
tail -20 $O/pmc_traffic.out | cut -c1-200
cp gpurun_out/pmc/traffic_FETCH_SIZE.txt gpurun_out/pmc/traffic_WRITE_SIZE.txt $O/ 2>/dev/null
( time python bench.py ) > $O/bench_default.log 2>&1
grep '^{' $O/bench_default.log | tail -1 | cut -c1-600
cd /tmp; rm -rf /tmp/tr
( time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/bench.py --no-cpu-baseline ) > $R/$O/prof_bench_default.log 2>&1
cd $R
T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); S=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
cp "$S" $O/bench_kernel_stats.csv 2>/dev/null
python tools/step_from_trace.py "$T" --top 70 --by-grid > $O/step_trace.txt 2>&1
head -40 $O/step_trace.txt | cut -c1-160
echo done

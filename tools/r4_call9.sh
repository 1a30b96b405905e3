#!/bin/bash
# Session 9: (a) capture crash narrowed: all-to-all fork_joins and the real HRNet encoder, (b) kernel traces (csv) of the HRNet
# step and of configs[4], (c) the configs[4] line with its roofline object.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp
O=gpurun_out/r4c9; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline"
( time timeout 600 python tools/capture_fork_min.py ) > $O/fork_min.log 2>&1
grep "^==" $O/fork_min.log
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hrtrace -o hr -- python $R/bench.py --encoder hrnet32 $B ) > $O/hr_trace.log 2>&1
cp $(find /tmp/hrtrace -name "*kernel_stats.csv" | head -1) $O/hrnet_kernel_stats.csv
grep '^{' $O/hr_trace.log | cut -c1-200
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5trace -o c5 -- python $R/bench.py --config5 --steps 10 --warmup 3 --no-roofline ) > $O/c5_trace.log 2>&1
cp $(find /tmp/c5trace -name "*kernel_stats.csv" | head -1) $O/config5_kernel_stats.csv
grep '^{' $O/c5_trace.log | cut -c1-200
( time timeout 600 python bench.py --config5 ) > $O/config5.log 2>&1
grep '^{' $O/config5.log | cut -c1-1500
( timeout 600 python -m pytest tests/test_bench_spawn.py -q -m gpu -x ) > $O/pytest_config5.log 2>&1
tail -3 $O/pytest_config5.log
echo done

#!/bin/bash
# Round 5, GPU session 4: loss-value diagnostic v3; BatchNorm kernels against a copy; kernel trace of the default bench command with
# the list of launches that are not kernels of this package (torch glue).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c4; mkdir -p $O
R=$(pwd)
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
run diag_ex0_s0 python tools/r5_diag_hr.py --exchange 0 --side 0 --reps 5
run diag_ex0_s0_resnet python tools/r5_diag_hr.py --exchange 0 --side 0 --reps 5 --encoder resnet50
grep -h "config\|replay\|losses" $O/diag_*.log | grep -v " 0 of" | cut -c1-500
run bn_bench python tools/bn_bench.py
grep -v "amdgpu.ids" $O/bn_bench.log | cut -c1-220
cd /tmp; rm -rf /tmp/tr
( time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/bench.py --no-cpu-baseline --no-reference-loop --steps 12 --warmup 3 ) > $R/$O/prof_bench.log 2>&1
cd $R
T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); S=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
cp "$S" $O/bench_kernel_stats.csv 2>/dev/null
python tools/step_from_trace.py "$T" --top 70 --torch > $O/step_trace.txt 2>&1
head -75 $O/step_trace.txt | cut -c1-170
sed -n '/launches that are not/,$p' $O/step_trace.txt | cut -c1-230
grep '^{' $O/prof_bench.log | tail -1 | cut -c1-300
echo done

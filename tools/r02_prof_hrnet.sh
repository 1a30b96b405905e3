#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r02_hrnet
mkdir -p "$OUT"
cd /tmp
rm -rf /tmp/prof_hr
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_hr -o hr -- python $R/bench.py --encoder hrnet32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-loop ) > $R/$OUT/prof.log 2>&1
cp /tmp/prof_hr/hr_kernel_stats.csv $R/$OUT/hrnet_kernel_stats.csv
cp /tmp/prof_hr/hr_kernel_trace.csv $R/$OUT/hrnet_kernel_trace.csv
cd $R
python bench.py --encoder hrnet32 --steps 5 --warmup 2 --no-cpu-baseline --no-reference-loop --dump-gemm $OUT/hrnet_gemm_profile.json > $OUT/bench.log 2>&1
tail -1 $OUT/bench.log | cut -c1-200

#!/bin/bash
# Headline evidence after the TrainStep guard fix (tools/r02_final3.sh ran with the batched reductions switched off by a
# misfiring shared-parameter check): TrainStep-dependent benches + rocprofv3 kernel summary / trace of the headline command.
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r02_final4
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
run smoke       python __graft_entry__.py smoke
run bench       python bench.py --dump-gemm "$OUT/gemm_profile.json"
run bench_b     python bench.py --family b --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_bmano python bench.py --family b-mano --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_hrnet python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_dist1 python bench.py --steps 10 --warmup 3 --force-dist --no-cpu-baseline --no-roofline --no-reference-loop
cd /tmp
rm -rf /tmp/prof_step
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-loop ) > $R/$OUT/prof_bench.log 2>&1
cp /tmp/prof_step/step_kernel_stats.csv $R/$OUT/bench_kernel_stats.csv
cp /tmp/prof_step/step_kernel_trace.csv $R/$OUT/bench_kernel_trace.csv
cd $R
echo done

#!/bin/bash
# End-of-round evidence of the committed state (after the statistics epilogue, the 17..32-column planner rule, the hand-major
# MANO form and the 16-byte softmax): full GPU suite, benches, rocprofv3 kernel summary / trace of the headline command.
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r02_final6
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
T=1500 run pytest_gpu python -m pytest tests -q -m gpu
run smoke       python __graft_entry__.py smoke
run bench       python bench.py --dump-gemm "$OUT/gemm_profile.json"
run bench_b     python bench.py --family b --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_bmano python bench.py --family b-mano --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_hrnet python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_dist1 python bench.py --steps 10 --warmup 3 --force-dist --no-cpu-baseline --no-roofline --no-reference-loop
run mano_bench  python tools/mano_bench.py --hands 128 4096 --json "$OUT/mano_bench.json"
cd /tmp
rm -rf /tmp/prof_step
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-loop ) > $R/$OUT/prof_bench.log 2>&1
cp /tmp/prof_step/step_kernel_stats.csv $R/$OUT/bench_kernel_stats.csv
cp /tmp/prof_step/step_kernel_trace.csv $R/$OUT/bench_kernel_trace.csv
cd $R
echo done

#!/bin/bash
# Round 4, GPU session 2: engine 2 with the bound atomics fixed (one candidate per block, read-before-atomic).
#   1. e2_bench with the error figures taken in the PRODUCTION configuration (same tile, same split-K chunk)
#   2. training step engine 1 vs engine 2, same box, live GEMM profile
#   3. rocprofv3 kernel trace of the engine-2 step (one replay cut out by tools/step_from_trace.py)
#   4. the whole GPU suite with RIH_GEMM_ENGINE=2
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r4c2
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop"
run e2_bench python tools/e2_bench.py
run train_e1 python bench.py $Q --dump-gemm "$OUT/gemm_profile_e1.json"
run train_e2 env RIH_GEMM_ENGINE=2 python bench.py $Q --dump-gemm "$OUT/gemm_profile_e2.json"
run train_e1b python bench.py $Q --no-roofline
run train_e2b env RIH_GEMM_ENGINE=2 python bench.py $Q --no-roofline
( cd /tmp && rm -rf /tmp/prof_e2 && RIH_GEMM_ENGINE=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e2 -o step -- \
    python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-loop --no-roofline ) > "$OUT/prof_e2.log" 2>&1
f=$(find /tmp/prof_e2 -name "*kernel_stats.csv" | head -1); cp "$f" "$OUT/bench_kernel_stats_e2.csv" 2>/dev/null
f=$(find /tmp/prof_e2 -name "*kernel_trace.csv" | head -1)
python tools/step_from_trace.py "$f" --top 70 > "$OUT/step_trace_e2.txt" 2>&1; head -n 30 "$OUT/step_trace_e2.txt"
run hrnet_e1 python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
run hrnet_e2 env RIH_GEMM_ENGINE=2 python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
T=1500 run pytest_gpu_e2 env RIH_GEMM_ENGINE=2 python -m pytest tests -q -m gpu -x
echo done

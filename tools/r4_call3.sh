#!/bin/bash
# Round 4, GPU session 3: engine 2 with bound BLOCKS (64 atomic lines per bound) and the pipelined main loop (two LDS stages, one
# barrier per k-tile, conversion interleaved with the MFMAs) against the two-barrier loop (librenderih_amd_nopipe.so) and engine 1.
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r4c3
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop"
NP=$R/renderih_amd/librenderih_amd_nopipe.so
run e2_bench_pipe python tools/e2_bench.py
run e2_bench_nopipe env RIH_AB_LIB=$NP python tools/e2_bench.py
run train_e1 python bench.py $Q --no-roofline
run train_e2_pipe env RIH_GEMM_ENGINE=2 python bench.py $Q --dump-gemm "$OUT/gemm_profile_e2_pipe.json"
run train_e2_nopipe env RIH_GEMM_ENGINE=2 RIH_AB_LIB=$NP python bench.py $Q --no-roofline
run train_e1b python bench.py $Q --no-roofline
run train_e2_pipe_b env RIH_GEMM_ENGINE=2 python bench.py $Q --no-roofline
RIH_PMC_ENGINE=2 bash tools/gemm_pmc.sh r4c3/gemm_pmc_e2_pipe > "$OUT/gemm_pmc_e2_pipe.log" 2>&1; tail -n 13 "$OUT/gemm_pmc_e2_pipe.log"
( cd /tmp && rm -rf /tmp/prof_e2 && RIH_GEMM_ENGINE=2 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e2 -o step -- \
    python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-reference-loop --no-roofline ) > "$OUT/prof_e2.log" 2>&1
f=$(find /tmp/prof_e2 -name "*kernel_stats.csv" | head -1); cp "$f" "$OUT/bench_kernel_stats_e2.csv" 2>/dev/null
f=$(find /tmp/prof_e2 -name "*kernel_trace.csv" | head -1)
python tools/step_from_trace.py "$f" --top 70 > "$OUT/step_trace_e2.txt" 2>&1; head -n 24 "$OUT/step_trace_e2.txt"
run hrnet_e1 python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
run hrnet_e2 env RIH_GEMM_ENGINE=2 python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
T=900 run pytest_e2 env RIH_GEMM_ENGINE=2 python -m pytest tests -q -m gpu -x -k "model_eval_matches or model_train_matches or conv2d or batchnorm or conv_bn or b64 or bench_shapes or hipgraph or train_step or grouped or gemm or linear or dropout or fp64"
echo done

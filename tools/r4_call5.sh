#!/bin/bash
# Round 4, GPU session 5: engine 2's 256x128 software-pipelined kernel (tile 4): per shape (PMC) and in the planner (RIH_E2_TILE4=1).
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r4c5
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
T=600 run pytest_tile4 python -m pytest tests -q -m gpu -x -k "tile4"
RIH_PMC_ENGINE=2 RIH_PMC_TILE4=1 bash tools/gemm_pmc.sh r4c5/gemm_pmc_e2_tile4 > "$OUT/gemm_pmc_e2_tile4.log" 2>&1; tail -n 13 "$OUT/gemm_pmc_e2_tile4.log"
run train_base python bench.py $Q
run train_tile4 env RIH_E2_TILE4=1 python bench.py $Q
run train_tile4_k128 env RIH_E2_TILE4=1 RIH_E2_TILE4_MINK=128 python bench.py $Q
run train_tile4_min256 env RIH_E2_TILE4=1 RIH_E2_TILE4_MIN=256 python bench.py $Q
run train_base_b python bench.py $Q
T=900 run pytest_tile4_model env RIH_E2_TILE4=1 python -m pytest tests -q -m gpu -x -k "model_eval_matches or model_train_matches or conv2d or conv_bn or b64 or bench_shapes or train_step"
echo done

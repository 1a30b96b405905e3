#!/bin/bash
# Round 4, GPU session 1 (one box, ~25 min):
#   1. engine 2 (three fp16 MFMA products) against engine 1 and rocBLAS fp32 on the twelve hottest shapes: time, TF, error vs fp64
#   2. its PMC pass (clock, MFMA-busy) on the same shapes
#   3. the training step with RIH_GEMM_ENGINE=2 against the default, same box, with the live GEMM profile
#   4. GPU parity tests of the network with RIH_GEMM_ENGINE=2
#   5. the four opt-ins that round 3 built but never ran on a GPU (tools/pending_ab.sh, condensed)
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r4c1
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop"
run e2_bench python tools/e2_bench.py
RIH_PMC_ENGINE=2 bash tools/gemm_pmc.sh r4c1/gemm_pmc_e2 > "$OUT/gemm_pmc_e2.log" 2>&1; tail -n 14 "$OUT/gemm_pmc_e2.log"
run train_e1 python bench.py $Q --dump-gemm "$OUT/gemm_profile_e1.json"
run train_e2 env RIH_GEMM_ENGINE=2 python bench.py $Q --dump-gemm "$OUT/gemm_profile_e2.json"
T=900 run pytest_e2 env RIH_GEMM_ENGINE=2 python -m pytest tests -q -m gpu -x -k "model_eval_matches or model_train_matches or conv2d or batchnorm or conv_bn or b64 or bench_shapes or hipgraph or train_step"
Q="$Q --no-roofline"
run pytest_dead_mid env RIH_SKIP_DEAD_MID=1 python -m pytest tests -q -m gpu -x -k "dead_mid or model_eval_matches or model_train_matches or fp16_backbone"
run train_skip_dead_mid env RIH_SKIP_DEAD_MID=1 python bench.py $Q
run train_t128 env RIH_WGRAD_GROUP_T128=128 python bench.py $Q
run pytest_bn_lastblock env RIH_BN_LASTBLOCK=1 python -m pytest tests -q -m gpu -x -k "batchnorm or conv_bn or model_train_matches or hrnet_eval or side_streams"
run train_bn_lastblock env RIH_BN_LASTBLOCK=1 python bench.py $Q
run pytest_gemm_dropout env RIH_GEMM_DROPOUT=1 python -m pytest tests -q -m gpu -x -k "linear_dropout or dropout or model_train_matches or hipgraph"
run train_gemm_dropout env RIH_GEMM_DROPOUT=1 python bench.py $Q
run train_base2 python bench.py $Q
run hrnet_base python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
run hrnet_bn_lastblock env RIH_BN_LASTBLOCK=1 python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
run hrnet_e2 env RIH_GEMM_ENGINE=2 python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
run config5_base python bench.py --config5
run config5_skip_dead_mid env RIH_SKIP_DEAD_MID=1 python bench.py --config5
echo done

#!/bin/bash
# Round 4, GPU session 4: engine 2 with the weight operands pre-split into two fp16 planes once per step (rih_presplit_multi,
# b_mode 2) against converting them in every GEMM loader (RIH_E2_PRESPLIT=0); where the GEMM's issue cycles go (second PMC pass).
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r4c4
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop"
NP=$R/renderih_amd/librenderih_amd_nopipe.so
run train_pre python bench.py $Q --dump-gemm "$OUT/gemm_profile_pre.json"
run train_nopre env RIH_E2_PRESPLIT=0 python bench.py $Q --no-roofline
run train_pre_nopipe env RIH_AB_LIB=$NP python bench.py $Q --no-roofline
run train_e1 env RIH_GEMM_ENGINE=1 python bench.py $Q --no-roofline
run train_pre_b python bench.py $Q --no-roofline
RIH_PMC_ENGINE=2 RIH_PMC_PRESPLIT=1 bash tools/gemm_pmc.sh r4c4/gemm_pmc_e2_pre > "$OUT/gemm_pmc_e2_pre.log" 2>&1; tail -n 13 "$OUT/gemm_pmc_e2_pre.log"
RIH_PMC_ENGINE=2 RIH_PMC_PRESPLIT=1 bash tools/gemm_pmc2.sh r4c4/gemm_pmc2_e2_pre > "$OUT/gemm_pmc2_e2_pre.log" 2>&1; tail -n 14 "$OUT/gemm_pmc2_e2_pre.log"
RIH_PMC_ENGINE=2 bash tools/gemm_pmc2.sh r4c4/gemm_pmc2_e2 > "$OUT/gemm_pmc2_e2.log" 2>&1; tail -n 14 "$OUT/gemm_pmc2_e2.log"
RIH_PMC_ENGINE=1 bash tools/gemm_pmc2.sh r4c4/gemm_pmc2_e1 > "$OUT/gemm_pmc2_e1.log" 2>&1; tail -n 14 "$OUT/gemm_pmc2_e1.log"
T=900 run pytest_pre python -m pytest tests -q -m gpu -x -k "model_eval_matches or model_train_matches or conv2d or batchnorm or conv_bn or b64 or bench_shapes or hipgraph or train_step or presplit or fp64 or weight_planes"
echo done

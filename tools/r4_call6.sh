#!/bin/bash
# Round 4, GPU session 6: epilogue with wave-level instead of workgroup barriers (A/B against librenderih_amd_blockbar.so);
# tile sweep of the forward / data-gradient shapes on engine 2.
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r4c6
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
BB=$R/renderih_amd/librenderih_amd_blockbar.so
run train_wavebar python bench.py $Q
run train_blockbar env RIH_AB_LIB=$BB python bench.py $Q
run train_wavebar_b python bench.py $Q
run train_blockbar_b env RIH_AB_LIB=$BB python bench.py $Q
run tile_sweep_e2 env RIH_SWEEP_ENGINE=2 RIH_SWEEP_WGRAD=0 python tools/tile_sweep.py
cat "$OUT/tile_sweep_e2.log" | grep "^fwd" | cut -c1-170
T=900 run pytest_epi python -m pytest tests -q -m gpu -x -k "conv2d or conv_bn or gemm or linear or grouped or bench_shapes or model_eval_matches or dropout"
echo done

#!/bin/bash
# last call of the round: statistics epilogue with the block's first row as the common shift (no per-lane Chan merges)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m39
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-120}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run pytest_ops python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_bn"
RIH_GEMM_STATS=0 run pass_1 $B
run epilogue_1 $B
T=100 run pytest_hrnet python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "hrnet_matches_fp64_oracle"
echo done

#!/bin/bash
# same-box A/B: split-K reductions behind every weight-gradient GEMM vs one batched launch per backward stage
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m20
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run pytest_ops python -m pytest tests/test_gpu_ops.py -x -q -m gpu
run pytest_trainstep python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "train_step or TrainStep or exchange"
for i in 1 2; do
RIH_DEFER_REDUCE=0 run immediate_$i $B
run deferred_$i $B
done
echo done

#!/bin/bash
# same-box A/B: Conv -> ReLU -> BN backward: ReLU gate folded into bn_bwd_apply vs a separate relu_bwd pass
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m35
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run pytest_ops python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_bn or batchnorm or conv2d"
for i in 1 2; do
RIH_RELU_GATE=0 run separate_$i $B
run folded_$i $B
done
echo done

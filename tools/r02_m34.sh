#!/bin/bash
# same-box A/B: BatchNorm statistics from the GEMM's 16-byte epilogue (rih_gemm_desc.stats) vs the separate statistics pass
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m34
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run pytest_ops python -m pytest tests/test_gpu_ops.py -x -q -m gpu
for i in 1 2; do
RIH_GEMM_STATS=0 run pass_$i $B
run epilogue_$i $B
done
echo done

#!/bin/bash
# same-box A/B: 16-byte softmax kernels vs the previous elementwise library
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m29
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
PREV=$PWD/renderih_amd/_ab/librenderih_amd_prev.so
run pytest_ops python -m pytest tests/test_gpu_ops.py -x -q -m gpu
for i in 1 2; do
RIH_AB_LIB=$PREV run prev_$i $B
run new_$i $B
done
echo done

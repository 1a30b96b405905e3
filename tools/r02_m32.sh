#!/bin/bash
# same-box A/B: 17..32-column GEMMs on the split engine (half-empty 64-wide tiles) vs the native-f32 128x32 tile
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m32
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
H="python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline"
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run pytest_ops python -m pytest tests/test_gpu_ops.py -x -q -m gpu
for i in 1 2; do
RIH_N32_SPLIT=0 run hr_old_$i $H
run hr_new_$i $H
done
RIH_N32_SPLIT=0 run rn_old $B
run rn_new $B
echo done

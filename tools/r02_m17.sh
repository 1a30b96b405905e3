#!/bin/bash
# same-box A/B: narrow (column-per-lane) vs wide (16-byte) GEMM epilogue, old vs new planner rules
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m17
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
NARROW=$PWD/renderih_amd/_ab/librenderih_amd_narrow.so
for i in 1 2; do
RIH_AB_LIB=$NARROW RIH_PLAN=1 run narrow_plan1_$i $B
RIH_PLAN=1 run wide_plan1_$i $B
run wide_plan2_$i $B
done
echo done

#!/bin/bash
# same-box A/B of the final (shifted-sum / Chan) statistics epilogue against the separate statistics pass
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m38
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-300}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
for i in 1 2; do
RIH_GEMM_STATS=0 run pass_$i $B
run epilogue_$i $B
done
echo done

#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m16
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 2 "$OUT/$name.log" | cut -c1-260; }

run bench_prof python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --dump-gemm "$OUT/gemm_profile.json"
run bench  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline
echo done

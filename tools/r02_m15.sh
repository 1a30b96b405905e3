#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m15
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 2 "$OUT/$name.log" | cut -c1-260; }

run pytest_ops python -m pytest tests/test_gpu_ops.py -x -q -m gpu
run tile_sweep python tools/tile_sweep.py
run bench  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline
echo done

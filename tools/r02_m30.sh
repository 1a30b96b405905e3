#!/bin/bash
# MANO fused forward: hand-chunk-major form (variant 2; what variant 0 picks from 2048 hands on) vs tile-major (variant 3)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m30
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 3 "$OUT/$name.log" | cut -c1-300; }
run pytest_mano python -m pytest tests/test_gpu_mano.py -x -q -m gpu
run tile_major python tools/mano_bench.py --hands 128 1024 2048 4096 16384 --variant 3
run hand_major python tools/mano_bench.py --hands 128 1024 2048 4096 16384 --variant 2
run auto python tools/mano_bench.py --hands 128 4096 --variant 0 --json "$OUT/mano_bench.json"
echo done

#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m13
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 2 "$OUT/$name.log" | cut -c1-300; }
run pytest_p3 python -m pytest tests/test_gpu_p3.py -x -q -m gpu
run p3_bench  python tools/p3_bench.py --json "$OUT/p3_bench.json"
echo done

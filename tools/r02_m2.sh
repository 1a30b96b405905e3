#!/bin/bash
# Round 2, second GPU call: first hardware contact of the P3 GEMM (parity tests, per-shape benchmark) + the regular suite.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m2
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-300}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 3 "$OUT/$name.log" | cut -c1-400; }
run pytest_p3  python -m pytest tests/test_gpu_p3.py -x -q -m gpu
run p3_bench   python tools/p3_bench.py --json "$OUT/p3_bench.json"
T=900 run pytest_gpu python -m pytest tests -x -q -m gpu
echo done

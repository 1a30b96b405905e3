#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m8
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 5 "$OUT/$name.log" | cut -c1-700; }
run pytest_mano   python -m pytest tests/test_gpu_mano.py -x -q -m gpu
run mano_phases   python tools/mano_phases.py
run mano_fused    python tools/mano_bench.py --hands 128 1024 4096 16384 --json "$OUT/mano_fused.json"
echo done

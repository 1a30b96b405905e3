#!/bin/bash
# Last check of the committed state: full GPU suite, smoke, headline bench (the state after the 16-byte softmax kernels).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_final5
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
T=1500 run pytest_gpu python -m pytest tests -q -m gpu
run smoke python __graft_entry__.py smoke
run bench python bench.py
echo done

#!/bin/bash
# Re-check of the committed state after the last kernel edits (P3S layout, ABI v5): GPU suite, smoke, headline bench.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_final2
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 2 "$OUT/$name.log" | cut -c1-300; }
T=1500 run pytest_gpu python -m pytest tests -q -m gpu
run smoke python __graft_entry__.py smoke
run bench python bench.py
echo done

#!/bin/bash
# model-level GPU tests + headline bench after the last change (ReLU gate in bn_bwd_apply)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_final7
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-900}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
run pytest_model python -m pytest tests/test_gpu_model.py tests/test_gpu_lijun.py -q -m gpu
run smoke python __graft_entry__.py smoke
run bench python bench.py
echo done

#!/bin/bash
# same-box A/B: torch fused Adam (24 launches) vs renderih_amd.optim.Adam (one launch)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m21
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run pytest_ops python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "adam or deferred"
for i in 1 2; do
run torch_adam_$i $B --torch-adam
run rih_adam_$i $B
done
echo done

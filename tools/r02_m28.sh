#!/bin/bash
# same-box A/B after the guard fix: immediate reductions / per-call packs vs the batched ones
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m28
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run pytest_trainstep python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "train_step"
for i in 1 2; do
RIH_PACK_CACHE=0 RIH_DEFER_REDUCE=0 run neither_$i $B
RIH_PACK_CACHE=0 run defer_only_$i $B
run both_$i $B
done
echo done

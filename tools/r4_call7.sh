#!/bin/bash
# Round 4, GPU session 7: the mid convolutions without their channel concatenation (ops.conv1x1_cat, RIH_CAT_FREE) and the
# engine-2 planner rule for short reductions (RIH_E2_SHORTK_TILE1), each against its switch on one box.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r4c7
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-300; }
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
T=900 run pytest_cat python -m pytest tests -q -m gpu -x -k "conv1x1_cat or model_eval_matches or model_train_matches or fp64 or b64 or train_step or dead_mid or hipgraph"
run train_new python bench.py $Q
run train_cat env RIH_CAT_FREE=0 python bench.py $Q
run train_t2 env RIH_E2_SHORTK_TILE1=0 python bench.py $Q
run train_old env RIH_CAT_FREE=0 RIH_E2_SHORTK_TILE1=0 python bench.py $Q
run train_new_b python bench.py $Q
echo done

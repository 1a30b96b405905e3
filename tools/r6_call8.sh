#!/bin/bash
# Round 6, GPU session 8: k-tile rotation per row tile in rows_kernel (L2 channel hot spot?) -- per shape and whole step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c8}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-160; }
for r in 0 1 3 5; do
T=300 run rows_bench_rot$r env RIH_ROWS_ROT=$r python tools/rows_bench.py
grep -E "fwd conv1 L3 |fwd conv3 L3|fwd conv1 L2 |dgrad ds L2|fwd conv1 L4.0|per training" $O/rows_bench_rot$r.log | cut -c1-150
done
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run base_a python bench.py $Q
run rot1_a env RIH_ROWS_ROT=1 python bench.py $Q
run base_b python bench.py $Q
run rot1_b env RIH_ROWS_ROT=1 python bench.py $Q
echo done

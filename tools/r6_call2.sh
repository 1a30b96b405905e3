#!/bin/bash
# Round 6, GPU session 2: rows_kernel with the two-k-tile A prefetch (raw LDS-DMA, counted waits) -- parity, per-shape, step A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c2}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
T=600 run pytest_rows python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "rows or panel or halo"
tail -3 $O/pytest_rows.log | cut -c1-200
T=600 run rows_bench python tools/rows_bench.py
cat $O/rows_bench.log | cut -c1-220
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run train_rows1 python bench.py $Q
run train_rows0 env RIH_ROWS=0 python bench.py $Q
run train_rows1_b python bench.py $Q
run train_rows0_b env RIH_ROWS=0 python bench.py $Q
echo done

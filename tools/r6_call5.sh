#!/bin/bash
# Round 6, GPU session 5: stem kernel, rows planning rule (ROWS_MIN_M), 128x64 grouped weight gradients -- parity, then same-box A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c5}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-160; }
T=600 run pytest_new python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "rows or stem or 128x64 or conv2d or grouped"
tail -3 $O/pytest_new.log | cut -c1-200
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
for rep in a b; do
run base_$rep python bench.py $Q
run stem0_$rep env RIH_STEM=0 python bench.py $Q
run wgt1_$rep env RIH_WGRAD_T1=1 python bench.py $Q
run rowsall_$rep env RIH_ROWS_MIN_M=1 python bench.py $Q
run rows0_$rep env RIH_ROWS=0 python bench.py $Q
done
echo done

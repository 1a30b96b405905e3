#!/bin/bash
# Round 6, GPU session 10: after the dead-register fix of rows_kernel's odd trip count -- the failed tests, the anomaly tool, the suite.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c10}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-220; }
T=600 run pytest_quick python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "rows or stem or conv2d"
tail -3 $O/pytest_quick.log | cut -c1-200
T=300 run reduce_anomaly python tools/r6_reduce_anomaly.py
grep -E "ok |WRONG|cases" $O/reduce_anomaly.log | cut -c1-330
T=2400 run pytest_gpu python -m pytest tests -q -m gpu
tail -14 $O/pytest_gpu.log | cut -c1-300
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run base_a python bench.py $Q
run rows0_a env RIH_ROWS=0 RIH_STEM=0 python bench.py $Q
run base_b python bench.py $Q
run rows0_b env RIH_ROWS=0 RIH_STEM=0 python bench.py $Q
echo done

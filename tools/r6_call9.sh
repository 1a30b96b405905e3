#!/bin/bash
# Round 6, GPU session 9: the whole GPU suite on the pruned tree with the tightened bars and the new tests; HRNet-W32 in four stages
# (plain, with the RCCL exchange at world 1); the in-graph reduction anomaly reduced.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c9}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-220; }
T=300 run reduce_anomaly python tools/r6_reduce_anomaly.py
grep -E "ok |WRONG|cases" $O/reduce_anomaly.log | cut -c1-400
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run hr_plain python bench.py --encoder hrnet32 $Q
run hr_dist python bench.py --encoder hrnet32 --force-dist $Q
grep -o '"comm_ms_exposed[^,]*\|"gradient_buckets_MB[^]]*\]' $O/hr_dist.log | head -4
run res_dist python bench.py --force-dist $Q
T=2400 run pytest_gpu python -m pytest tests -q -m gpu
tail -30 $O/pytest_gpu.log | cut -c1-300
echo done

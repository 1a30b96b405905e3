#!/bin/bash
# Round 6, GPU session 15: whole GPU suite on the vmcnt(0) library, then same-box A/B (rows + stem off).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c15}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
T=2400 run pytest_gpu python -m pytest tests -q -m gpu
tail -8 $O/pytest_gpu.log | cut -c1-300
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run base_a python bench.py $Q
run rows0_a env RIH_ROWS=0 RIH_STEM=0 python bench.py $Q
run base_b python bench.py $Q
run rows0_b env RIH_ROWS=0 RIH_STEM=0 python bench.py $Q
run hr_base python bench.py --encoder hrnet32 $Q
run hr_rows0 env RIH_ROWS=0 python bench.py --encoder hrnet32 $Q
echo done

#!/bin/bash
# Round 5, GPU session 11 (and 12, 13 ...): MANO backward -- per-hand kernel (scalar SE3 operands, dG on the MFMA), tile-major blend, finish kernel.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r5c11}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -4 | cut -c1-330; }
run pytest_mano python -m pytest tests/test_gpu_mano.py -q
tail -3 $O/pytest_mano.log
run mano_bench python tools/mano_bench.py --hands 128 1024 4096
run mano_phases python tools/mano_phases.py
tail -2 $O/mano_phases.log
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mano -o mano -- python $OLDPWD/tools/mano_bench.py --hands 128 4096 --iters 20 > /dev/null 2>&1 )
f=$(find /tmp/prof_mano -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $O/mano_kernel_stats.csv && grep -i mano $O/mano_kernel_stats.csv | cut -c1-60,200-400
t=$(find /tmp/prof_mano -name '*kernel_trace.csv' | head -1); [ -n "$t" ] && python tools/mano_trace_split.py "$t" > $O/mano_trace_split.txt 2>&1; cat $O/mano_trace_split.txt | tail -12
echo done

#!/bin/bash
# Last session of round 4: the whole GPU suite, smoke and the benches on the FINAL binary (after the evidence session the fp16
# convolution tile rule, the MANO layouts and the per-test gradient bands changed; the training path did not).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4final2; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-260; }
T=1500 run pytest_gpu python -m pytest tests -q -m gpu
grep "passed\|failed" $O/pytest_gpu.log | tail -2
run smoke python __graft_entry__.py smoke
run bench_final2 python bench.py
run config5 python bench.py --config5
run bench_bmano python bench.py --family b-mano --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
echo done

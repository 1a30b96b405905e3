#!/bin/bash
# Round 5, GPU session 10: stacked QKV operands through the pack cache -- parity (model tests that run TrainStep) and step A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c10; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
run pytest_ts python -m pytest tests/test_gpu_model.py -q -k "train_step or hipgraph or paired_decoder or family_b_train"
tail -3 $O/pytest_ts.log
RIH_STACK_CACHE=0 run train_stack0 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_STACK_CACHE=1 run train_stack1 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_STACK_CACHE=0 run train_stack0_b python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_STACK_CACHE=1 run train_stack1_b python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
echo done

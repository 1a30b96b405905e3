#!/bin/bash
# Round 6, evidence session part 3: the whole GPU suite and smoke() on the final tree.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6final3}; mkdir -p $O
( time timeout 2400 python -m pytest tests -q -m gpu -x ) > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -5 | cut -c1-200
( time timeout 600 python __graft_entry__.py smoke ) > $O/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/smoke.log | cut -c1-200
( timeout 600 python bench.py ) > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log | cut -c1-200
echo done

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c11}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; tail -3 $O/$n.log | cut -c1-250; }
run stem1 python tools/r6_diag_stem.py
run stem0 env RIH_STEM=0 python tools/r6_diag_stem.py
T=900 run mag_default python -m pytest tests/test_gpu_round5.py -q -m gpu -x -k "magnitude and frozen"
grep -E "replay on|passed|failed" $O/mag_default.log | cut -c1-200
T=900 run mag_stem0 env RIH_STEM=0 python -m pytest tests/test_gpu_round5.py -q -m gpu -x -k "magnitude and frozen"
grep -E "replay on|passed|failed" $O/mag_stem0.log | cut -c1-200
T=900 run mag_rows0 env RIH_ROWS=0 python -m pytest tests/test_gpu_round5.py -q -m gpu -x -k "magnitude and frozen"
grep -E "replay on|passed|failed" $O/mag_rows0.log | cut -c1-200
echo done

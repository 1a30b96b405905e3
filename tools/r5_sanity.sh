#!/bin/bash
# Round 5: the last library once more (after the evidence session only rih_mano.hip changed: forward blend staging, phase stamps).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5sanity; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -2 | cut -c1-260; }
run pytest_mano python -m pytest tests/test_gpu_mano.py tests/test_pose_head.py -q -m gpu
tail -2 $O/pytest_mano.log
run pytest_model_b python -m pytest tests/test_gpu_model.py -q -k "family_b or mano or new_model"
tail -2 $O/pytest_model_b.log
run smoke python __graft_entry__.py smoke
tail -2 $O/smoke.log | cut -c1-160
run mano_bench python tools/mano_bench.py --hands 128 1024 4096
run mano_phases python tools/mano_phases.py
grep -E "third|backward" $O/mano_phases.log | cut -c1-400
run bench_bmano python bench.py --family b-mano --no-cpu-baseline --no-reference-loop
echo done

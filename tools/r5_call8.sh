#!/bin/bash
# Round 5, GPU session 8: short-K streaming GEMM (panel kernel) -- parity, per-shape A/B, step A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c8; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
run pytest_panel python -m pytest tests/test_gpu_ops.py -q -k "panel or conv3x3_halo or conv2d"
tail -3 $O/pytest_panel.log
run panel_bench python tools/panel_bench.py
grep -v "^$\|amdgpu.ids" $O/panel_bench.log | cut -c1-220
RIH_PANEL=0 run train_panel0 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_PANEL=1 run train_panel1 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_PANEL=0 run train_panel0_b python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_PANEL=1 run train_panel1_b python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_PANEL=0 run hr_panel0 python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline
RIH_PANEL=1 run hr_panel1 python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline
echo done

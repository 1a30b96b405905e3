#!/bin/bash
# Round 5, GPU session 6: fp16 halo-resident 3x3 kernel (configs[4]) -- parity on the GPU, per-shape sweep and the configs[4] line
# with the kernel off / on; the revised magnitude test.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c6; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-260; }
run pytest_half python -m pytest tests/test_gpu_paths.py -q -k "fp16 or graphed or half"
tail -2 $O/pytest_half.log
RIH_HCONV_HALO=0 run sweep_halo0 python tools/hconv_sweep.py
RIH_HCONV_HALO=1 run sweep_halo1 python tools/hconv_sweep.py
grep " k3\|3x3\|total\|sum" $O/sweep_halo0.log | cut -c1-200 | head -20
echo ---
grep " k3\|3x3\|total\|sum" $O/sweep_halo1.log | cut -c1-200 | head -20
RIH_HCONV_HALO=0 run config5_halo0 python bench.py --config5 --no-cpu-baseline
RIH_HCONV_HALO=1 run config5_halo1 python bench.py --config5 --no-cpu-baseline
RIH_HCONV_HALO=0 run config5_halo0_b python bench.py --config5 --no-cpu-baseline --no-roofline
RIH_HCONV_HALO=1 run config5_halo1_b python bench.py --config5 --no-cpu-baseline --no-roofline
T=900 run pytest_mag python -m pytest tests/test_gpu_round5.py -q -s -k "magnitude"
grep -n "passed\|failed\|replay on" $O/pytest_mag.log | cut -c1-200
echo done

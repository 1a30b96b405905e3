#!/bin/bash
# Round 5, GPU session 18: the two new convolution kernels together, same box: RIH_HALO3=0 RIH_PANEL=0 against the default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c18; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
RIH_HALO3=0 RIH_PANEL=0 run train_r4kernels python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
run train_default python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_HALO3=0 RIH_PANEL=0 run train_r4kernels_b python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
run train_default_b python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_HALO3=0 RIH_PANEL=0 run hr_r4kernels python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline
run hr_default python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline
echo done

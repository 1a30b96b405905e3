#!/bin/bash
# Round 6, call 24: the halo kernel's residual epilogue (rih_conv3_desc.r, ABI 19): its tests, HRNet-W32 with RIH_HALO3_RES=0 / 1 in
# three interleaved pairs, one ResNet50 line, then the whole GPU suite and smoke.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6c24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "halo or residual_preconditions" > $O/pytest_halo.log 2>&1; echo "halo tests exit $?"; tail -3 $O/pytest_halo.log | cut -c1-200
for i in a b c; do for r in 0 1; do
  RIH_HALO3_RES=$r timeout 600 python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline > $O/hr_res${r}_$i.log 2>&1
  echo "hr halo3_res $r $i $(grep '^{' $O/hr_res${r}_$i.log | cut -c1-120)"
done; done
OUT=r6c24 bash tools/r6_final3.sh

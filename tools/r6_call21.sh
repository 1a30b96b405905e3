#!/bin/bash
# Round 6, call 21: the halo 3x3 kernel on HRNet's 16 x 16 branch (32 patches x 128 channels: 64 workgroups on 64-wide channel blocks)
# with 32-wide channel blocks (128 workgroups) when the 64-wide grid is below RIH_C3_BN64_MIN_WGS: per shape, then the HRNet step.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6c21; mkdir -p $O
for m in 0 128 256; do
  echo "== RIH_C3_BN64_MIN_WGS=$m"
  CONV3_SET=hrnet RIH_C3_BN64_MIN_WGS=$m timeout 300 python tools/conv3_bench.py 2>&1 | grep -v amdgpu.ids | tee $O/conv3_bench_hrnet_min$m.log | cut -c1-220
done
for i in a b; do for m in 0 128 256; do
  RIH_C3_BN64_MIN_WGS=$m timeout 600 python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline > $O/hr_min${m}_$i.log 2>&1
  echo "hr min64wgs $m $i $(grep '^{' $O/hr_min${m}_$i.log | cut -c1-120)"
done; done
for m in 0 256; do
  RIH_C3_BN64_MIN_WGS=$m timeout 600 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline > $O/rn_min${m}.log 2>&1
  echo "resnet min64wgs $m $(grep '^{' $O/rn_min${m}.log | cut -c1-120)"
done
echo done

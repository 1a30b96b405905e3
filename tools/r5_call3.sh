#!/bin/bash
# Round 5, GPU session 3: generalized halo kernel (16x16 patches, 32-channel blocks) -- parity, per-shape A/B (ResNet50 and HRNet-W32
# shapes), step A/B on both encoders; replay determinism diagnostic v2 (outputs too); the exchange + side streams test again.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c3; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
run pytest_halo python -m pytest tests/test_gpu_ops.py -q -k "conv3x3_halo or conv2d"
tail -3 $O/pytest_halo.log
run conv3_bench python tools/conv3_bench.py
grep -v "^$\|amdgpu.ids" $O/conv3_bench.log | cut -c1-220
CONV3_SET=hrnet run conv3_bench_hrnet python tools/conv3_bench.py
grep -v "^$\|amdgpu.ids" $O/conv3_bench_hrnet.log | cut -c1-220
RIH_HALO3=0 run train_halo0 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_HALO3=1 run train_halo1 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_HALO3=0 run hr_halo0 python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline
RIH_HALO3=1 run hr_halo1 python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline
run diag_ex1_s0 python tools/r5_diag_hr.py --exchange 1 --side 0 --reps 8
run diag_ex1_s0_b python tools/r5_diag_hr.py --exchange 1 --side 0 --reps 8
run diag_ex0_s0 python tools/r5_diag_hr.py --exchange 0 --side 0 --reps 8
run diag_ex1_s3 python tools/r5_diag_hr.py --exchange 1 --side 3 --reps 8
grep -h "config\|replay.*differ\|losses" $O/diag_*.log | grep -v " 0 of" | cut -c1-400
T=900 run pytest_r5_exch python -m pytest tests/test_gpu_round5.py -q -s -k "survive_the_gradient_exchange"
grep -n "passed\|failed\|worst relative\|differs" $O/pytest_r5_exch.log | cut -c1-300
echo done

#!/bin/bash
# Round 5, GPU session 2: halo-resident 3x3 convolution -- parity on the GPU, per-shape A/B against the implicit GEMM, step A/B;
# replay determinism of the captured HRNet step (diagnostic); HRNet gradient band on engine 0 (are the loose tensors ReLU flips?).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c2; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
run pytest_halo python -m pytest tests/test_gpu_ops.py -q -k "conv3x3_halo or conv2d or conv1x1_cat"
tail -3 $O/pytest_halo.log
run conv3_bench python tools/conv3_bench.py
cat $O/conv3_bench.log | cut -c1-220
RIH_HALO3=0 run train_halo0 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_HALO3=1 run train_halo1 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_HALO3=0 run train_halo0_b python bench.py --no-cpu-baseline --no-reference-loop --no-roofline
RIH_HALO3=1 run train_halo1_b python bench.py --no-cpu-baseline --no-reference-loop
run diag_ex1_s3 python tools/r5_diag_hr.py --exchange 1 --side 3
run diag_ex0_s3 python tools/r5_diag_hr.py --exchange 0 --side 3 --reps 8
run diag_ex1_s0 python tools/r5_diag_hr.py --exchange 1 --side 0
run diag_ex1_s3_g0 python tools/r5_diag_hr.py --exchange 1 --side 3 --group 0
grep -h "config\|replay\|losses" $O/diag_*.log | cut -c1-300
RIH_GEMM_ENGINE=0 T=400 run pytest_hr_e0 python -m pytest tests/test_gpu_model.py -q -s -k "hrnet_matches_fp64_oracle and False"
grep -n "passed\|failed\|_grad_report" $O/pytest_hr_e0.log | cut -c1-200
echo done

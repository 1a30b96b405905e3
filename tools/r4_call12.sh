#!/bin/bash
# Session 12: two k-tiles in flight for short plain reductions on engine 2 (RIH_E2_DEEP_K), same-box A/B + the GEMM tests;
# the fp64-anchored gradient reports printed (for the band of tests/test_gpu_model.py::_grad_report).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4c12; mkdir -p $O
B="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run() { n=$1; shift; echo "== $n: $*"; ( time timeout 600 "$@" ) > $O/$n.log 2>&1; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
run train_deep256 python bench.py $B
run train_deep0 env RIH_E2_DEEP_K=0 python bench.py $B
run train_deep1024 env RIH_E2_DEEP_K=1024 python bench.py $B
run train_deep4096 env RIH_E2_DEEP_K=4096 python bench.py $B
run train_deep256_b python bench.py $B
run train_deep0_b env RIH_E2_DEEP_K=0 python bench.py $B
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_shapes.py -q -m gpu -x ) > $O/pytest_ops.log 2>&1
tail -2 $O/pytest_ops.log
( timeout 900 python -m pytest tests/test_gpu_model.py -q -m gpu -x -rP -k "fp64_oracle or b64 or train_matches" ) > $O/pytest_grad_reports.log 2>&1
grep "_grad_report\|passed\|failed" $O/pytest_grad_reports.log | cut -c1-250
echo done

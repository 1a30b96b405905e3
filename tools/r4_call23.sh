#!/bin/bash
# Session 23: the reduction pass of the BatchNorm backward folded into the data-gradient GEMM's epilogue (RIH_BN_FOLD=1):
# same-box A/B first, then its GPU tests and the model-level gradient tests with the fold on.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4c23; mkdir -p $O
B="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
run() { n=$1; shift; echo "== $n: $*"; ( time timeout 300 "$@" ) > $O/$n.log 2>&1; grep '^{' $O/$n.log | tail -1 | cut -c1-200; tail -3 $O/$n.log | grep -i "error\|Traceback" ; }
run train_base python bench.py $B
run train_fold env RIH_BN_FOLD=1 python bench.py $B
run train_base_b python bench.py $B
run train_fold_b env RIH_BN_FOLD=1 python bench.py $B
( timeout 200 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "bn_backward_sums or batchnorm" ) > $O/pytest_fold_ops.log 2>&1; tail -2 $O/pytest_fold_ops.log
( timeout 170 env RIH_BN_FOLD=1 python -m pytest tests/test_gpu_model.py tests/test_gpu_bench_shapes.py -q -m gpu -x -k "model_matches_fp64_oracle or b64 or train_step_graph or family_b_matches" ) > $O/pytest_fold_model.log 2>&1; tail -2 $O/pytest_fold_model.log
echo done

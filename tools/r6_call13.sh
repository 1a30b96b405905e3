#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
t() { echo "== $*"; env "$@" 2>&1 | tail -2 | cut -c1-160; }
t A=1 timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -k "conv_problem or deterministic"
t A=1 timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -k "linear_problem or deterministic"
t AMD_SERIALIZE_KERNEL=3 timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -k "conv_problem or deterministic"
t RIH_ROWS_MINK=512 timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -k "conv_problem or deterministic"
t RIH_ROWS_MIN_M=65536 timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -k "conv_problem or deterministic"

#!/bin/bash
# Session 14: MANO hand-major kernels with the next half tile in flight: GPU tests + micro-benchmark (before: 126.9 / 501 us).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4c14; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_mano.py -q -m gpu -x ) > $O/pytest_mano.log 2>&1
tail -2 $O/pytest_mano.log
( timeout 300 python tools/mano_bench.py --hands 128 1024 4096 16384 --json $O/mano_bench.json ) > $O/mano_bench.log 2>&1
grep '^{' $O/mano_bench.log | cut -c1-300
( timeout 300 python tools/mano_bench.py --hands 4096 --json $O/mano_bench_b.json ) > $O/mano_bench_b.log 2>&1
grep '^{' $O/mano_bench_b.log | cut -c1-300
echo done

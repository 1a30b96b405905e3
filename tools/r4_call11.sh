#!/bin/bash
# Session 11: fp16 backbone convolutions with the residual / epilogue constants requested before the k-loop.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4c11; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_paths.py tests/test_bench_spawn.py -q -m gpu -x ) > $O/pytest_fp16.log 2>&1
tail -3 $O/pytest_fp16.log
( timeout 300 python tools/hconv_sweep.py 256 ) > $O/hconv_sweep.log 2>&1
head -40 $O/hconv_sweep.log | cut -c1-160
( timeout 300 env RIH_HCONV_GLDS=0 python tools/hconv_sweep.py 256 ) > $O/hconv_sweep_regs.log 2>&1
head -3 $O/hconv_sweep_regs.log | cut -c1-160
( timeout 600 python bench.py --config5 ) > $O/config5.log 2>&1
grep '^{' $O/config5.log | cut -c1-400
echo done

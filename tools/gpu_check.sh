#!/bin/bash
# One GPU-box call: full GPU test suite (no -x, so every failure is reported), smoke, bench.
# Everything of interest is written under gpurun_out/ (merged back into the repo by gpurun).
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 >> gpurun_out/env.log
nproc >> gpurun_out/env.log; lscpu | grep "Model name" >> gpurun_out/env.log
timeout ${T_TEST:-900} python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
if [ "${SKIP_BENCH}" != "1" ]; then
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 --dump-gemm gpurun_out/gemm_profile.json > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log; tail -3 gpurun_out/bench.log
fi
if [ "${PROFILE}" = "1" ]; then
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1
cd $GRAFT_REPO_ROOT; ls -la gpurun_out/prof | head; find gpurun_out/prof -name "*stats*" | head
fi

#!/bin/bash
# One GPU-box call: full GPU test suite (no -x, so every failure is reported), smoke, bench.
# Everything of interest is written under gpurun_out/ (merged back into the repo by gpurun).
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch;print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 >> gpurun_out/env.log
nproc >> gpurun_out/env.log; lscpu | grep "Model name" >> gpurun_out/env.log
if [ "${SKIP_TESTS}" != "1" ]; then
timeout ${T_TEST:-900} python -m pytest ${PYTEST_ARGS:-tests} -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
fi
if [ "${SKIP_BENCH}" != "1" ]; then
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 --dump-gemm gpurun_out/gemm_profile.json > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log; tail -3 gpurun_out/bench.log
fi
if [ "${PROFILE}" = "1" ]; then
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/prof.log 2>&1
cd $R; mkdir -p gpurun_out/prof
find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/prof/ \;
find /tmp/prof -type f | head -20 >> gpurun_out/prof.log
ls -la gpurun_out/prof
fi
if [ "${GEMMBENCH}" = "1" ]; then
timeout 600 python tools/gemm_bench.py > gpurun_out/gemm_bench.log 2>&1; tail -3 gpurun_out/gemm_bench.log
fi
if [ "${PMC}" = "1" ]; then
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc -o pmc -- python $R/tools/gemm_bench.py --quick > $R/gpurun_out/pmc.log 2>&1
cd $R; mkdir -p gpurun_out/pmc; find /tmp/pmc -name "*counter_collection*.csv" -exec cp {} gpurun_out/pmc/ \; ; ls -la gpurun_out/pmc; find /tmp/pmc -type f | head >> gpurun_out/pmc.log
fi

#!/bin/bash
# PMC passes over tools/gemm_bench.py --quick (counters only with --kernel-trace; one rocprofv3 run per counter set).
# Output: gpurun_out/pmc/<set>_counter_collection.csv
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc$i -o pmc -- python $R/tools/gemm_bench.py ${PMC_MODE:---quick} > $R/gpurun_out/pmc/run$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection*.csv" | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/pmc/set${i}_counter_collection.csv
done
ls -la $R/gpurun_out/pmc

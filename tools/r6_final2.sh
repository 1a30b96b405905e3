#!/bin/bash
# Round 6, evidence session part 2 (after the no-GC-during-capture fix of TrainStep): the whole GPU suite, PMC traffic.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6final2}; mkdir -p $O
( time timeout 2400 python -m pytest tests -q -m gpu ) > $O/pytest_gpu.log 2>&1; echo "pytest exit $?"; grep -E "passed|failed|^FAILED|^ERROR" $O/pytest_gpu.log | tail -8 | cut -c1-200
bash tools/pmc_traffic.sh > $O/pmc_traffic_stdout.txt 2>&1
cp gpurun_out/pmc/traffic_FETCH_SIZE.txt gpurun_out/pmc/traffic_WRITE_SIZE.txt $O/ 2>/dev/null
head -3 $O/traffic_FETCH_SIZE.txt | cut -c1-200; head -3 $O/traffic_WRITE_SIZE.txt | cut -c1-200
echo done

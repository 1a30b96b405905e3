#!/bin/bash
# Round 6, evidence session: whole GPU suite, smoke, the default bench line, the kernel trace of that same command, the other
# configurations, PMC traffic of the GEMM family, PMC table of rows_kernel -- all on one box, on the library of the last commit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd)
O=gpurun_out/${OUT:-r6final}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-260; }
if [ -z "$SKIP_TESTS" ]; then
T=2400 run pytest_gpu python -m pytest tests -q -m gpu -x
tail -4 $O/pytest_gpu.log | cut -c1-200
fi
run smoke python __graft_entry__.py smoke
tail -2 $O/smoke.log | cut -c1-200
run bench_default python bench.py
cd /tmp; rm -rf /tmp/tr
( time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/bench.py --no-cpu-baseline ) > $R/$O/prof_bench_default.log 2>&1
cd $R
T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); S=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
cp "$S" $O/bench_kernel_stats.csv 2>/dev/null
python tools/step_from_trace.py "$T" --top 70 --by-grid > $O/step_trace.txt 2>&1
head -6 $O/step_trace.txt | cut -c1-160
unset T
run bench_hrnet python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop
run bench_hrnet_dist1 python bench.py --encoder hrnet32 --force-dist --no-cpu-baseline --no-reference-loop
run bench_b python bench.py --family b --no-cpu-baseline --no-reference-loop
run bench_bmano python bench.py --family b-mano --no-cpu-baseline --no-reference-loop
run config5 python bench.py --config5
run bench_dist1 python bench.py --force-dist --no-cpu-baseline --no-reference-loop
run mano_bench python tools/mano_bench.py --hands 128 4096 --json $O/mano_bench.json
run rows_bench python tools/rows_bench.py
bash tools/r5_pmc.sh $(basename $O)/pmc_rows 'rows_kernel|gemm_split_kernel' tools/rows_pmc_driver.py > $O/pmc_rows_stdout.txt 2>&1
cp $O/pmc_rows/pmc_table.txt $O/pmc_table_rows.txt 2>/dev/null; cat $O/pmc_table_rows.txt | cut -c1-200
bash tools/pmc_traffic.sh > $O/pmc_traffic_stdout.txt 2>&1
cp gpurun_out/pmc/traffic_FETCH_SIZE.txt gpurun_out/pmc/traffic_WRITE_SIZE.txt $O/ 2>/dev/null
head -2 $O/traffic_FETCH_SIZE.txt; head -2 $O/traffic_WRITE_SIZE.txt
echo done

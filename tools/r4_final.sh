#!/bin/bash
# Final evidence session of round 4: the whole GPU suite, smoke, PMC traffic (-> profiles/r04/traffic_v5.json, read by bench.py),
# the default bench line, the kernel trace of the SAME default command, the other families / configurations.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
export TMPDIR=/tmp GRAFT_REPO_ROOT=$R
O=gpurun_out/r4final; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-260; }
T=1500 run pytest_gpu python -m pytest tests -q -m gpu
tail -3 $O/pytest_gpu.log
run smoke python __graft_entry__.py smoke
bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1
cp gpurun_out/pmc/traffic_FETCH_SIZE.txt $O/pmc_traffic_FETCH_SIZE_final.txt; cp gpurun_out/pmc/traffic_WRITE_SIZE.txt $O/pmc_traffic_WRITE_SIZE_final.txt
python tools/make_traffic_json.py gpurun_out/pmc profiles/r04/traffic_v5.json "raw files pmc_traffic_*_final.txt; engine 2 default, concatenation-free mid convolutions" > $O/traffic_v5.log 2>&1
cp profiles/r04/traffic_v5.json $O/traffic_v5.json
run bench_final python bench.py --dump-gemm $O/gemm_profile_final.json
( cd /tmp && rm -rf /tmp/prof_final && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -o step -- python $R/bench.py ) > $O/prof_bench_default_command.log 2>&1
cp /tmp/prof_final/step_kernel_stats.csv $O/bench_kernel_stats_final.csv
python tools/step_from_trace.py /tmp/prof_final/step_kernel_trace.csv > $O/step_trace_final.txt 2>&1
grep '^{' $O/prof_bench_default_command.log | tail -1 | cut -c1-200
run bench_hrnet python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop
run bench_b python bench.py --family b --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_bmano python bench.py --family b-mano --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run config5 python bench.py --config5
run mano python tools/mano_bench.py --hands 128 4096 --json $O/mano_bench.json
run dist1 python bench.py --steps 10 --warmup 3 --force-dist --no-cpu-baseline --no-reference-loop --no-roofline
echo done

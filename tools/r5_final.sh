#!/bin/bash
# Round 5, evidence session: whole GPU suite, smoke, the default bench line, the kernel trace of that same command, the other
# configurations -- all on one box, on the library of the last commit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$(pwd)
O=gpurun_out/${OUT:-r5final}; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-900} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-260; }
T=1500 run pytest_gpu python -m pytest tests -q -m gpu -x
tail -4 $O/pytest_gpu.log | cut -c1-200
run smoke python __graft_entry__.py smoke
tail -2 $O/smoke.log | cut -c1-200
run bench_default python bench.py
cd /tmp; rm -rf /tmp/tr
( time timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- python $R/bench.py --no-cpu-baseline ) > $R/$O/prof_bench_default.log 2>&1
cd $R
T=$(find /tmp/tr -name "*kernel_trace.csv" | head -1); S=$(find /tmp/tr -name "*kernel_stats.csv" | head -1)
cp "$S" $O/bench_kernel_stats.csv 2>/dev/null
python tools/step_from_trace.py "$T" --top 70 > $O/step_trace.txt 2>&1
head -6 $O/step_trace.txt | cut -c1-160
unset T
run bench_hrnet python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop
run bench_b python bench.py --family b --no-cpu-baseline --no-reference-loop
run bench_bmano python bench.py --family b-mano --no-cpu-baseline --no-reference-loop
run config5 python bench.py --config5
run bench_dist1 python bench.py --force-dist --no-cpu-baseline --no-reference-loop
run mano_bench python tools/mano_bench.py --hands 128 4096 --json $O/mano_bench.json
echo done

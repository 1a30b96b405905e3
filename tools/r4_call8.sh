#!/bin/bash
# Session 8: (a) HRNet kernel trace, (b) multi-side-stream capture crash: minimal cases, then the real step under faulthandler
# and rocgdb, (c) weight-gradient group chunk under engine 2, (d) configs[4] kernel trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4c8; mkdir -p $O
B="--steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline"
run() { n=$1; shift; echo "== $n: $*"; ( time timeout 400 "$@" ) > $O/$n.log 2>&1; tail -2 $O/$n.log | cut -c1-300; }
run fork_min python tools/capture_fork_min.py
grep "^==" $O/fork_min.log
run hr_fault env RIH_SIDE_STREAMS=2 RIH_SIDE_CAPTURE_MAX=2 python -X faulthandler bench.py --encoder hrnet32 $B
grep -n "Fatal\|File \|Segmentation" $O/hr_fault.log | head -30
run hr_gdb env RIH_SIDE_STREAMS=2 RIH_SIDE_CAPTURE_MAX=2 timeout 300 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint" -ex run -ex bt -ex "info sharedlibrary" --args python bench.py --encoder hrnet32 $B
grep -n "^#[0-9]" $O/hr_gdb.log | head -40
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/hrtrace -o hr -- python $OLDPWD/bench.py --encoder hrnet32 $B ) > $O/hr_trace.log 2>&1
find /tmp/hrtrace -name "*kernel_stats.csv" -exec cp {} $O/hrnet_kernel_stats.csv \;
tail -1 $O/hr_trace.log | cut -c1-300
run wg512 env RIH_WGRAD_GROUP_KCHUNK=512 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline
run wg2048 env RIH_WGRAD_GROUP_KCHUNK=2048 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline
run wgdef python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/c5trace -o c5 -- python $OLDPWD/bench.py --config5 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline ) > $O/c5_trace.log 2>&1
find /tmp/c5trace -name "*kernel_stats.csv" -exec cp {} $O/config5_kernel_stats.csv \;
tail -1 $O/c5_trace.log | cut -c1-300
echo done

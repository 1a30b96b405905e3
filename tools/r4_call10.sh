#!/bin/bash
# Session 10: (a) the capture crash with the autograd hop in fork_join (and, once, without: RIH_FORK_HOP=0), (b) HRNet step with
# 1 / 2 / 3 side streams under capture, (c) HRNet GPU tests, (d) per-shape timing of the fp16 backbone's convolutions.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4c10; mkdir -p $O
B="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
( time timeout 600 python tools/capture_fork_min.py ) > $O/fork_min_hop.log 2>&1
grep "^==" $O/fork_min_hop.log
( RIH_FORK_HOP=0 timeout 120 python -X faulthandler tools/capture_fork_min.py a2a2 ) > $O/fork_min_nohop_a2a2.log 2>&1; echo "a2a2 without the hop: rc $?"
run() { n=$1; shift; echo "== $n: $*"; ( time timeout 400 "$@" ) > $O/$n.log 2>&1; grep '^{' $O/$n.log | cut -c1-200; tail -4 $O/$n.log | grep -i "error\|fault\|dumped" ; }
run hr_side1 python bench.py --encoder hrnet32 $B
run hr_side2 env RIH_SIDE_STREAMS=2 RIH_SIDE_CAPTURE_MAX=2 python -X faulthandler bench.py --encoder hrnet32 $B
run hr_side3 env RIH_SIDE_STREAMS=3 RIH_SIDE_CAPTURE_MAX=3 python -X faulthandler bench.py --encoder hrnet32 $B
run hr_side0 env RIH_SIDE_STREAMS=0 python bench.py --encoder hrnet32 $B
run hr_side1_b python bench.py --encoder hrnet32 $B
( timeout 900 env RIH_SIDE_STREAMS=3 RIH_SIDE_CAPTURE_MAX=3 python -m pytest tests -q -m gpu -x -k "hrnet or side_stream or fork" ) > $O/pytest_hrnet_side3.log 2>&1
tail -3 $O/pytest_hrnet_side3.log
( timeout 300 python tools/hconv_sweep.py 256 ) > $O/hconv_sweep.log 2>&1
head -40 $O/hconv_sweep.log | cut -c1-160
echo done

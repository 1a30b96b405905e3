#!/bin/bash
# Round 6, GPU session 17: RIH_WGRAD_T1 (128x64 grouped weight gradients for <= 64 output channels), three interleaved pairs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c17}; mkdir -p $O
Q="--steps 30 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
for r in a b c; do
( python bench.py $Q ) > $O/t1off_$r.log 2>&1; grep '^{' $O/t1off_$r.log | cut -c1-130
( RIH_WGRAD_T1=1 python bench.py $Q ) > $O/t1on_$r.log 2>&1; grep '^{' $O/t1on_$r.log | cut -c1-130
done
( python bench.py --encoder hrnet32 $Q ) > $O/hr_t1off.log 2>&1; grep '^{' $O/hr_t1off.log | cut -c1-130
( RIH_WGRAD_T1=1 python bench.py --encoder hrnet32 $Q ) > $O/hr_t1on.log 2>&1; grep '^{' $O/hr_t1on.log | cut -c1-130
echo done

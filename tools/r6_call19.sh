#!/bin/bash
# Round 6, call 19: ROWS_MIN_N = 128 (layer1's N = 64 launches back on the tiled 128 x 64 kernel) against 64, three interleaved pairs of
# the step; then the whole GPU suite and smoke on that default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6c19; mkdir -p $O
for i in a b c; do for n in 64 128; do
  RIH_ROWS_MIN_N=$n timeout 600 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline > $O/train_minn${n}_$i.log 2>&1
  echo "min_n $n $i $(grep '^{' $O/train_minn${n}_$i.log | cut -c1-140)"
done; done
OUT=r6c19 bash tools/r6_final3.sh

#!/bin/bash
# Round 6, call 18: rows_kernel on N = 64 (HBM-bound layer1 shapes, the stem aside): 256 x 64 tiles (88 KB LDS: one workgroup per CU)
# against 128 x 64 tiles (56 KB: two per CU).  Per-shape microbenchmark, then the step, interleaved.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r6c18; mkdir -p $O
for bm in 256 128 256 128; do
  RIH_ROWS_N64_BM=$bm timeout 300 python tools/rows_bench.py 2>&1 | grep -E " N   64 " | sed "s/^/bm$bm  /" | tee -a $O/rows_bench_n64.log | cut -c1-170
done
for i in a b; do for bm in 256 128; do
  RIH_ROWS_N64_BM=$bm timeout 600 python bench.py --no-cpu-baseline --no-reference-loop --no-roofline > $O/train_bm${bm}_$i.log 2>&1
  echo "bm$bm $i $(grep '^{' $O/train_bm${bm}_$i.log | cut -c1-140)"
done; done
echo done

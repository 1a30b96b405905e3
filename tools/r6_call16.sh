#!/bin/bash
# Round 6, GPU session 16: eight partial sums in the split-K reduction against four (same box), the re-conditioned magnitude test.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c16}; mkdir -p $O
C=$PWD/renderih_amd/librenderih_amd_reduce4.so
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
for r in a b c; do
( python bench.py $Q ) > $O/train_reduce8_$r.log 2>&1; grep '^{' $O/train_reduce8_$r.log | cut -c1-130
( RIH_AB_LIB=$C python bench.py $Q ) > $O/train_reduce4_$r.log 2>&1; grep '^{' $O/train_reduce4_$r.log | cut -c1-130
done
( python bench.py --encoder hrnet32 $Q ) > $O/hr_reduce8.log 2>&1; grep '^{' $O/hr_reduce8.log | cut -c1-130
( RIH_AB_LIB=$C python bench.py --encoder hrnet32 $Q ) > $O/hr_reduce4.log 2>&1; grep '^{' $O/hr_reduce4.log | cut -c1-130
timeout 900 python -m pytest tests/test_gpu_round5.py -q -m gpu -k "magnitude" 2>&1 | grep -E "replay on|passed|failed" | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -m gpu -k "grouped or deferred or staged or wgrad" 2>&1 | tail -2
echo done

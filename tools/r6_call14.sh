#!/bin/bash
# Round 6, GPU session 14: counted waits (vmcnt(NPB + NPA)) against vmcnt(0) in rows_kernel -- reproducibility stress, per-shape
# speed, step A/B.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/${OUT:-r6c14}; mkdir -p $O
C=$PWD/renderih_amd/librenderih_amd_counted.so
( REPS=100 RIH_AB_LIB=$C python tools/r6_stress_rows.py ) > $O/stress_counted.log 2>&1; grep -E "differ|TOTAL" $O/stress_counted.log | cut -c1-170
( REPS=100 python tools/r6_stress_rows.py ) > $O/stress_wait0.log 2>&1; grep -E "TOTAL" $O/stress_wait0.log
( python tools/rows_bench.py ) > $O/rows_bench_wait0.log 2>&1; tail -1 $O/rows_bench_wait0.log
( RIH_AB_LIB=$C python tools/rows_bench.py ) > $O/rows_bench_counted.log 2>&1; tail -1 $O/rows_bench_counted.log
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline"
for r in a b; do
( python bench.py $Q ) > $O/train_wait0_$r.log 2>&1; grep '^{' $O/train_wait0_$r.log | cut -c1-130
( RIH_AB_LIB=$C python bench.py $Q ) > $O/train_counted_$r.log 2>&1; grep '^{' $O/train_counted_$r.log | cut -c1-130
done
timeout 600 python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -k "conv_problem or deterministic" 2>&1 | tail -2
echo done

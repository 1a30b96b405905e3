#!/bin/bash
# Session 22: the reference's unmodified eager loop on this package: engine 2 against engine 1, and a host-side profile.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4c22; mkdir -p $O
Q="--steps 3 --warmup 2 --no-cpu-baseline --no-roofline"
( timeout 300 python bench.py $Q ) > $O/eager_e2.log 2>&1; grep -o '"eager_reference_loop": {[^}]*}' $O/eager_e2.log | cut -c1-120
( timeout 300 env RIH_GEMM_ENGINE=1 python bench.py $Q ) > $O/eager_e1.log 2>&1; grep -o '"eager_reference_loop": {[^}]*}' $O/eager_e1.log | cut -c1-120
( timeout 300 env RIH_PROFILE_REF_LOOP=1 python bench.py $Q ) > $O/eager_e2_profile.log 2>&1; grep -A50 "tottime" $O/eager_e2_profile.log | head -60 | cut -c1-150
echo done

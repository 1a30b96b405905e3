#!/bin/bash
# Session 13: fp16 convolutions on 128x64 tiles for every Cout (RIH_HCONV_BN=64: three workgroups per CU) against the default.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r4c13; mkdir -p $O
( timeout 300 python tools/hconv_sweep.py 256 ) > $O/hconv_sweep_default.log 2>&1
head -3 $O/hconv_sweep_default.log | cut -c1-160
( timeout 300 env RIH_HCONV_BN=64 python tools/hconv_sweep.py 256 ) > $O/hconv_sweep_bn64.log 2>&1
head -40 $O/hconv_sweep_bn64.log | cut -c1-160
( timeout 600 python bench.py --config5 --no-roofline ) > $O/config5_default.log 2>&1
grep '^{' $O/config5_default.log | cut -c1-200
( timeout 600 env RIH_HCONV_BN=64 python bench.py --config5 --no-roofline ) > $O/config5_bn64.log 2>&1
grep '^{' $O/config5_bn64.log | cut -c1-200
echo done

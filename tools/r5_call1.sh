#!/bin/bash
# Round 5, GPU session 1: the new parity tests, the baseline bench line with the honest roofline fraction, HRNet with / without the
# gradient exchange (side-stream clamp lifted), the decoder-on-engine-2 potential (RIH_DECODER_E2=1: per-launch profile only).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c1; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-400; }
T=1500 run pytest_r5 python -m pytest tests/test_gpu_round5.py -q -s -x
grep -n "passed\|failed\|replay on\|gradient tensors\|configs\[0\]\|outside\|worst" $O/pytest_r5.log | cut -c1-260 | tail -60
T=900 run pytest_grads python -m pytest tests/test_gpu_model.py -q -s -k "hrnet_matches_fp64_oracle"
grep -n "passed\|failed\|_grad_report\|outside the" $O/pytest_grads.log | cut -c1-260 | tail -80
run bench_base python bench.py --no-cpu-baseline --no-reference-loop
run hr_plain python bench.py --encoder hrnet32 --no-cpu-baseline --no-reference-loop --no-roofline
run hr_dist python bench.py --encoder hrnet32 --force-dist --no-cpu-baseline --no-reference-loop --no-roofline
RIH_DP_SIDE_LIMIT=0 run hr_dist_clamped python bench.py --encoder hrnet32 --force-dist --no-cpu-baseline --no-reference-loop --no-roofline
RIH_DECODER_E2=1 run bench_decoder_e2 python bench.py --no-cpu-baseline --no-reference-loop --steps 5 --warmup 2
python - <<'PY'
import json,glob
for f in ('gpurun_out/r5c1/bench_base.log','gpurun_out/r5c1/bench_decoder_e2.log'):
    for line in open(f):
        if line.startswith('{'):
            d=json.loads(line); r=d['roofline']
            print(f, d['value'], 'img/s; gemm ms', r['gemm_ms_per_step'], 'frac', r['frac'], 'peak', r['peak'])
            for k,v in r['by_engine'].items(): print('   ',k,v)
PY
echo done

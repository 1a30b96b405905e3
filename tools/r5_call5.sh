#!/bin/bash
# Round 5, GPU session 5: the whole GPU suite on the current library (halo 3x3 on by default, ABI 15), smoke, the per-launch GEMM
# dump of one step (which plain-row 1x1 convolutions are how far from streaming speed).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/r5c5; mkdir -p $O
run() { n=$1; shift; echo "== $n: $*"; ( time timeout ${T:-600} "$@" ) > $O/$n.log 2>&1; echo "   exit $?"; grep '^{' $O/$n.log | tail -1 | cut -c1-200; }
T=1500 run pytest_gpu python -m pytest tests -q -m gpu
grep "passed\|failed" $O/pytest_gpu.log | tail -3
grep -n "^FAILED\|^ERROR" $O/pytest_gpu.log | head -20
run smoke python __graft_entry__.py smoke
tail -2 $O/smoke.log
run bench_dump python bench.py --no-cpu-baseline --no-reference-loop --dump-gemm $O/gemm_dump.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5c5/gemm_dump.json'))
cols=d['columns']; rows=d['rows']
print('plain-row launches (a_mode 0, K <= 512) with >= 30 us: us, TF/s, M N K tile, GB/s of algorithmic bytes (A + C; B negligible)')
for r in rows:
    us, tf, M, N, K, batch, am, bm, tile, sk = r[:10]
    if am == 0 and isinstance(tile, int) and tile in (0,1,2) and K and K <= 512 and us >= 30:
        by = 4.0*M*(K+N)*batch
        print('%7.1f us %6.1f TF  M %7d N %4d K %4d tile %d  %5.0f GB/s' % (us, tf, M, N, K, tile, by/us/1e3))
PY
echo done

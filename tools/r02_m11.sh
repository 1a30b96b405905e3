#!/bin/bash
# (historical: A/B of the BatchNorm statistics in the rih_gemm epilogue, RIH_FUSE_BN_STATS -- the code path was removed after this
# measurement; the script is kept as the record of what profiles/r02/bench_m11_*.log ran)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m11
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 2 "$OUT/$name.log" | cut -c1-260; }
run pytest_ops   python -m pytest tests/test_gpu_ops.py -x -q -m gpu
run bench_fused  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop
run bench_nostat env RIH_FUSE_BN_STATS=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline
T=900 run pytest_model python -m pytest tests/test_gpu_model.py -x -q -m gpu
echo done

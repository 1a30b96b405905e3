#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m6
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 4 "$OUT/$name.log" | cut -c1-700; }
run pytest_mano   python -m pytest tests/test_gpu_mano.py -x -q -m gpu
run mano_fused    python tools/mano_bench.py --hands 128 1024 4096 16384 --json "$OUT/mano_fused.json"
run mano_twokern  python tools/mano_bench.py --hands 128 4096 --variant 1 --json "$OUT/mano_two_kernel.json"
run pytest_shapes python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu -k "b16 or deterministic" -s
T=300 run prof_mano rocprofv3 --kernel-trace --stats -d "$OUT/prof_mano" -o mano -- python tools/mano_bench.py --hands 4096 --iters 20
find "$OUT/prof_mano" -name '*kernel_stats*' -exec cp {} "$OUT/mano_kernel_stats.csv" \; 2>/dev/null
ls -R "$OUT/prof_mano" | head -20
echo done

#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m5
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 4 "$OUT/$name.log" | cut -c1-700; }
run pytest_trainstep python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "train_step"
run bench            python bench.py --steps 20 --warmup 5
run bench_dist1      python bench.py --steps 10 --warmup 3 --force-dist --no-cpu-baseline --no-roofline --no-reference-loop
T=900 run pytest_shapes python -m pytest tests/test_gpu_bench_shapes.py -q -m gpu
T=900 run pytest_paths  python -m pytest tests/test_gpu_paths.py -q -m gpu
echo done

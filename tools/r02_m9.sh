#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m9
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 3 "$OUT/$name.log" | cut -c1-400; }
run pytest_trainstep python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "train_step"
run bench_side    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-reference-loop
run bench_noside  env RIH_SIDE_WGRAD=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-reference-loop
run bench_side_nograph    python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-reference-loop --no-graph
run bench_noside_nograph  env RIH_SIDE_WGRAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-reference-loop --no-graph
echo done

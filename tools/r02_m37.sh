#!/bin/bash
# Chan-merged statistics epilogue (robust to |mean| >> std) + HRNet / second family through encoder.conv_bn
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m37
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-200; }
run pytest_ops python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "conv_bn or batchnorm"
run pytest_variants python -m pytest tests/test_gpu_model.py -q -m gpu -k "hrnet or family_b or new_model or train_mode or gradient"
run bench python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-reference-loop --no-roofline
run hrnet python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-roofline
echo done

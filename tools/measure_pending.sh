#!/bin/bash
# One gpurun call that gives every path written after round 1's GPU budget was spent its first hardware numbers:
#   gpurun --timeout 1800 -- 'bash tools/measure_pending.sh'      (about 20 GPU-minutes)
# Writes gpurun_out/pending/*.log|json (copy what is to be judged into profiles/rNN/).  Every step has its own timeout; a
# failing step does not stop the following ones.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pending
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 3 "$OUT/$name.log"; }

T=1200 run pytest_pending python -m pytest tests/test_zz_gpu_pending.py -q -m gpu -rxXs
run bench_default   python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run bench_presplit1 env RIH_PRESPLIT=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_presplit2 env RIH_PRESPLIT=2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_fusedattn env RIH_FUSED_ATTN=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run infer_f32       python tools/infer_bench.py --iters 10
run infer_f32_fold  env RIH_FOLD_BN=1 python tools/infer_bench.py --iters 10
run infer_f16       python tools/infer_bench.py --iters 10 --fp16
run infer_f16_b64   python tools/infer_bench.py --iters 10 --fp16 --batch 64
run hconv_layers    python tools/hconv_bench.py --iters 10
run hconv_layers_regstage env RIH_HCONV_GLDS=0 python tools/hconv_bench.py --iters 10
T=900 run prof_infer_f16 rocprofv3 --kernel-trace --stats -d "$OUT/prof_infer_f16" -- python tools/infer_bench.py --iters 5 --fp16
find "$OUT/prof_infer_f16" -name '*kernel_stats*.csv' -exec cp {} "$OUT/infer_f16_kernel_stats.csv" \; 2>/dev/null
echo done

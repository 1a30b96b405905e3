#!/bin/bash
# Round 2, first GPU call: numbers for the paths that had none (pre-split GEMM operands per shape, fused attention A/B,
# fp16 inference = BASELINE configs[4], MANO micro-benchmark), plus a fresh default bench line.
#   gpurun --timeout 1500 -- 'bash tools/r02_m1.sh'
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/r02_m1
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-300}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 2 "$OUT/$name.log" | cut -c1-600; }

run presplit_bench  python tools/presplit_bench.py
run mano_bench      python tools/mano_bench.py --json "$OUT/mano_bench.json"
run bench_default   python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run bench_presplit1 env RIH_PRESPLIT=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_presplit2 env RIH_PRESPLIT=2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run bench_fusedattn env RIH_FUSED_ATTN=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline
run infer_f16       python tools/infer_bench.py --iters 10 --fp16
run infer_f32       python tools/infer_bench.py --iters 10
run hconv_layers    python tools/hconv_bench.py --iters 10
T=400 run prof_infer_f16 rocprofv3 --kernel-trace --stats -d "$OUT/prof_infer_f16" -- python tools/infer_bench.py --iters 5 --fp16
find "$OUT/prof_infer_f16" -name '*kernel_stats*.csv' -exec cp {} "$OUT/infer_f16_kernel_stats.csv" \; 2>/dev/null
T=300 run prof_mano rocprofv3 --kernel-trace --stats -d "$OUT/prof_mano" -- python tools/mano_bench.py --iters 20
find "$OUT/prof_mano" -name '*kernel_stats*.csv' -exec cp {} "$OUT/mano_kernel_stats.csv" \; 2>/dev/null
rm -rf "$OUT/prof_infer_f16" "$OUT/prof_mano"
echo done

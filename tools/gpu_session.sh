#!/bin/bash
# One GPU-box session: run a list of named steps, each with its own timeout, logs under gpurun_out/<tag>/.
#   tools/gpu_session.sh <tag> <step> [<step> ...]
# Steps: pytest[:<-k expr>]  pytestn[:<workers>] (whole GPU suite, pytest-xdist)  pytestf:<file>  smoke  bench[:<extra args>]  ab:<ENV=VAL>[,<ENV=VAL>...]  hrab:<ENV=VAL>[,...]  prof  mano  hrnet  famb  dist1  infer
# `ab:` runs bench.py twice in the SAME process environment apart from the given variables (baseline first), for same-box A/B.
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 1 "$OUT/$name.log" | cut -c1-400; }
QUICK="--no-cpu-baseline --no-reference-loop"
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  case $kind in
    pytest) if [ -n "$arg" ]; then T=1500 run "pytest_$(echo "$arg" | tr -c 'A-Za-z0-9' _)" python -m pytest tests -q -m gpu -x -k "$arg"; else T=1800 run pytest_gpu python -m pytest tests -q -m gpu; fi ;;
    pytestf) T=1500 run "pytestf_$(echo "$arg" | tr -c 'A-Za-z0-9' _)" python -m pytest $arg -q -m gpu -x ;;
    # (measured: 6 xdist workers on ONE GPU are SLOWER than the serial suite -- 141 of 295 tests in 540 s against 410-460 s for
    # all of them -- the oracle's fp64 CPU runs oversubscribe the host threads; kept for boxes with more than one GPU)
    pytestn) T=1500 run "pytest_gpu_n${arg:-6}" python -m pytest tests -q -m gpu -n "${arg:-6}" ;;
    smoke) run smoke python __graft_entry__.py smoke ;;
    bench) run "bench$(echo "$arg" | tr -c 'A-Za-z0-9' _)" python bench.py --dump-gemm "$OUT/gemm_profile.json" $arg ;;
    ab) n=0; for v in base $(echo "$arg" | tr ',' ' '); do
          n=$((n+1))
          if [ "$v" = base ]; then run "ab${n}_base" python bench.py --steps 20 --warmup 5 $QUICK --no-roofline
          else run "ab${n}_$(echo "$v" | tr -c 'A-Za-z0-9' _)" env $(echo "$v" | tr '+' ' ') python bench.py --steps 20 --warmup 5 $QUICK --no-roofline; fi
        done ;;
    abroof) for v in base $(echo "$arg" | tr ',' ' '); do
          if [ "$v" = base ]; then run "abroof_base" python bench.py --steps 20 --warmup 5 $QUICK --dump-gemm "$OUT/gemm_base.json"
          else run "abroof_$(echo "$v" | tr -c 'A-Za-z0-9' _)" env $(echo "$v" | tr '+' ' ') python bench.py --steps 20 --warmup 5 $QUICK --dump-gemm "$OUT/gemm_$(echo "$v" | tr -c 'A-Za-z0-9' _).json"; fi
        done ;;
    famb) run bench_b python bench.py --family b --steps 10 --warmup 3 $QUICK; run bench_bmano python bench.py --family b-mano --steps 10 --warmup 3 $QUICK ;;
    hrnet) run bench_hrnet python bench.py --encoder hrnet32 --steps 10 --warmup 3 $QUICK ;;
    hrab) for v in base $(echo "$arg" | tr ',' ' '); do
          if [ "$v" = base ]; then run "hrab_base" python bench.py --encoder hrnet32 --steps 10 --warmup 3 $QUICK
          else run "hrab_$(echo "$v" | tr -c 'A-Za-z0-9' _)" env $(echo "$v" | tr '+' ' ') python bench.py --encoder hrnet32 --steps 10 --warmup 3 $QUICK; fi
        done ;;
    dist1) run bench_dist1 python bench.py --steps 10 --warmup 3 --force-dist $QUICK --no-roofline ;;
    mano) run mano_bench python tools/mano_bench.py --hands 128 4096 --json "$OUT/mano_bench.json" ;;
    infer) run infer_f16 python tools/infer_bench.py --fp16; run infer_f32 python tools/infer_bench.py ;;
    prof) cd /tmp; rm -rf /tmp/prof_step
          ( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python $R/bench.py --steps 5 --warmup 2 $QUICK --no-roofline $arg ) > $R/$OUT/prof_bench.log 2>&1
          cp /tmp/prof_step/step_kernel_stats.csv $R/$OUT/bench_kernel_stats.csv 2>/dev/null
          cp /tmp/prof_step/step_kernel_trace.csv $R/$OUT/bench_kernel_trace.csv 2>/dev/null
          cd $R; python tools/step_from_trace.py $OUT/bench_kernel_trace.csv > $OUT/step_trace.txt 2>&1 ;;
    *) echo "unknown step $step" ;;
  esac
done
echo done

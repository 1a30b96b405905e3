#!/bin/bash
# Round-2 evidence set on one box: full GPU test-suite, headline bench (+ rocprofv3 kernel summary of the same command), HBM
# traffic PMC passes (separate runs, counters only), secondary configurations (second family, MANO-in-forward family,
# HRNet-W32, fp16 / fp32 inference = BASELINE configs[4]) and the MANO micro-benchmark.
cd "$(dirname "$0")/.." || exit 1
R=$(pwd)
export TMPDIR=/tmp
OUT=gpurun_out/r02_final3
mkdir -p "$OUT"
run() { name=$1; shift; echo "== $name: $*"; ( timeout "${T:-600}" "$@" ) > "$OUT/$name.log" 2>&1; echo "   exit $?"; tail -n 2 "$OUT/$name.log" | cut -c1-300; }
T=1500 run pytest_gpu python -m pytest tests -q -m gpu
run smoke       python __graft_entry__.py smoke
run bench       python bench.py --dump-gemm "$OUT/gemm_profile.json"
run bench_b     python bench.py --family b --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_bmano python bench.py --family b-mano --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_hrnet python bench.py --encoder hrnet32 --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop
run bench_dist1 python bench.py --steps 10 --warmup 3 --force-dist --no-cpu-baseline --no-roofline --no-reference-loop
run infer_f16   python tools/infer_bench.py --iters 10 --fp16
run infer_f32   python tools/infer_bench.py --iters 10
run mano_bench  python tools/mano_bench.py --hands 128 4096 --json "$OUT/mano_bench.json"
cd /tmp
rm -rf /tmp/prof_step /tmp/prof_inf /tmp/prof_mano
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_step -o step -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-reference-loop ) > $R/$OUT/prof_bench.log 2>&1
cp /tmp/prof_step/step_kernel_stats.csv $R/$OUT/bench_kernel_stats.csv
cp /tmp/prof_step/step_kernel_trace.csv $R/$OUT/bench_kernel_trace.csv
( timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inf -o inf -- python $R/tools/infer_bench.py --iters 5 --fp16 ) > $R/$OUT/prof_infer_f16.log 2>&1
cp /tmp/prof_inf/inf_kernel_stats.csv $R/$OUT/infer_f16_kernel_stats.csv
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_mano -o mano -- python $R/tools/mano_bench.py --hands 4096 --iters 20 ) > $R/$OUT/prof_mano.log 2>&1
cp /tmp/prof_mano/mano_kernel_stats.csv $R/$OUT/mano_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  ( timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-reference-loop ) > $R/$OUT/traffic_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection*.csv" | head -1)
  python - "$f" "$c" <<'PY' > $R/$OUT/pmc_traffic_$c.txt
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != c:
        continue
    k = r['Kernel_Name']
    name = k[k.index('gemm'):k.index('>') + 1] if 'gemm_' in k else ('other: ' + k.split('(')[0][-60:])
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += float(r['Counter_Value'])
tot_g = sum(v[1] for k, v in agg.items() if k.startswith('gemm'))
n_g = sum(v[0] for k, v in agg.items() if k.startswith('gemm'))
print('%s (raw counter units as reported by rocprofv3; 3 eager bench steps incl. warm-up)' % c)
print('rih_gemm kernels: launches %d total %.6g per-launch %.6g' % (n_g, tot_g, tot_g / max(n_g, 1)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print('%12.6g  n=%5d  per-launch %10.5g  %s' % (v[1], v[0], v[1] / v[0], k))
PY
  head -3 $R/$OUT/pmc_traffic_$c.txt
done
cd $R
echo done

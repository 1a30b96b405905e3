#!/bin/bash
# HBM traffic of the rih_gemm kernels during bench steps: two separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a
# pass on gfx950, MI355X_MICROARCH.md "rocprofv3 PMC slots"); counters only with --kernel-trace.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-graph --no-reference-loop > $R/gpurun_out/pmc/traffic_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection*.csv" | head -1)
  python - "$f" "$c" <<'PY' > $R/gpurun_out/pmc/traffic_$c.txt
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] != c:
        continue
    k = r['Kernel_Name']
    if 'maxpool_fwd' in k:
        steps = globals().get('steps', 0) + 1       # one launch per step: the number of steps the collection saw
    def upto(i0):       # the kernel's name from position i0 up to the end of its template arguments (or its parameter list)
        j = k.find('>', i0)
        j = j if j >= 0 else (k.find('(', i0) if k.find('(', i0) >= 0 else len(k)) - 1
        return k[i0:j + 1]
    if 'gemm_' in k:
        name = upto(k.index('gemm'))
    elif 'conv3x3_halo_kernel' in k or 'panel_kernel' in k or 'rows_kernel' in k:   # rounds 5-6: halo 3x3, streaming 1x1, long-K rows / stem
        i0 = k.index('conv3x3_halo') if 'conv3x3_halo' in k else k.index('panel_kernel') if 'panel_kernel' in k else k.index('rows_kernel')
        name = 'gemm-family ' + upto(i0)
    elif 'flash_' in k:
        name = 'gemm-family ' + upto(k.index('flash'))
    else:
        name = 'other: ' + k.split('(')[0][-60:]
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += float(r['Counter_Value'])
tot_g = sum(v[1] for k, v in agg.items() if k.startswith('gemm'))
n_g = sum(v[0] for k, v in agg.items() if k.startswith('gemm'))
steps = globals().get('steps', 0)
print('%s (raw counter units as reported by rocprofv3); steps seen: %d' % (c, steps))
print('GEMM family (rih_gemm, grouped launch, flash attention): launches %d total %.6g per-launch %.6g per-step %.6g'
      % (n_g, tot_g, tot_g / max(n_g, 1), tot_g / max(steps, 1)))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print('%12.6g  n=%5d  per-launch %10.5g  %s' % (v[1], v[0], v[1] / v[0], k))
PY
  tail -30 $R/gpurun_out/pmc/traffic_$c.txt | head -8
done

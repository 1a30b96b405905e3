#!/usr/bin/env python
"""Cut ONE hipGraph replay of the training step out of a rocprofv3 kernel trace (CSV) and summarise it: per-kernel launches,
total / average duration, and the idle time between consecutive kernels (the step boundary is a kernel launched once per step:
maxpool_fwd for the ResNet trunk, --mark NAME otherwise).
    python tools/step_from_trace.py trace.csv [--top 60] [--mark nchw_to_nhwc]"""
import collections
import csv
import re
import sys

MARK = 'maxpool_fwd'


def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'^void ', '', n)
    return n[:86]


def main():
    path = sys.argv[1]
    top = int(sys.argv[sys.argv.index('--top') + 1]) if '--top' in sys.argv else 60
    global MARK
    MARK = sys.argv[sys.argv.index('--mark') + 1] if '--mark' in sys.argv else 'maxpool_fwd'
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    marks = [i for i, r in enumerate(rows) if MARK in r["Kernel_Name"]]
    # the median-length one among the complete steps of the shortest kind: a hipGraph replay (the default bench command also
    # runs eager steps -- warm-up, the reference's loop, the per-launch profile -- which take longer; --last: the last step)
    if '--last' in sys.argv or len(marks) < 4:
        a, b = marks[-2], marks[-1]
    else:
        spans = sorted((int(rows[marks[i + 1]]['Start_Timestamp']) - int(rows[marks[i]]['Start_Timestamp']), i)
                       for i in range(len(marks) - 1))
        fast = [x for x in spans if x[0] <= 1.05 * spans[0][0]]
        i = fast[len(fast) // 2][1]
        a, b = marks[i], marks[i + 1]
        print('steps in the trace: %d; %d within 5 %% of the shortest (%.3f ms); shown: step %d' % (len(spans), len(fast), spans[0][0] / 1e6, i))
    step = rows[a:b]
    t0, t1 = int(step[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
    agg = collections.OrderedDict()
    busy, gaps, prev_end = 0, 0, None
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        k = agg.setdefault(short(r['Kernel_Name']), [0, 0])
        k[0] += 1
        k[1] += e - s
        busy += e - s
        if prev_end is not None and s > prev_end:
            gaps += s - prev_end
        prev_end = max(prev_end or e, e)
    print('step: %d launches, wall %.3f ms, sum of kernel durations %.3f ms, idle between kernels %.3f ms'
          % (len(step), (t1 - t0) / 1e6, busy / 1e6, gaps / 1e6))
    fam = ('gemm', 'conv3x3_halo', 'panel_kernel', 'rows_kernel', 'stem_kernel', 'wgrad')   # the GEMM family: rih_gemm's kernels, halo 3x3, panel / rows 1x1, stem
    g = sum(v[1] for k, v in agg.items() if k.startswith(fam))
    print('GEMM-family kernels (gemm_*, conv3x3_halo_kernel, panel_kernel, rows_kernel, stem_kernel) %.3f ms, everything else %.3f ms' % (g / 1e6, (busy - g) / 1e6))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print('%8.3f ms %5d x %8.1f us  %s' % (v[1] / 1e6, v[0], v[1] / v[0] / 1e3, k))
    if '--by-grid' in sys.argv:
        # the GEMM-family launches of the step by (kernel, grid): a shape is a grid size, so per-shape durations inside the replay
        byg = collections.OrderedDict()
        for r in step:
            n = short(r['Kernel_Name'])
            if not n.startswith(('gemm', 'conv3x3', 'panel_kernel', 'rows_kernel', 'stem_kernel', 'wgrad')):
                continue
            k = byg.setdefault((n[:70], r.get('Grid_Size_X', r.get('Grid_Size', '?'))), [0, 0])
            k[0] += 1
            k[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        print('GEMM-family launches by (kernel, grid threads):')
        for (n, g), v in sorted(byg.items(), key=lambda kv: -kv[1][1]):
            print('%8.3f ms %4d x %8.1f us  grid %9s  %s' % (v[1] / 1e6, v[0], v[1] / v[0] / 1e3, g, n))
    if '--torch' in sys.argv:
        # every launch of the step that is not a kernel of this package (torch element-wise glue, fills, copies), in launch
        # order with its grid size and the package kernels before / after it: enough to find the line of Python that issued it
        ours = ('gemm', 'bn_', 'flash', 'layernorm', 'cheby', 'dropout', 'relu_', 'splitk', 'absmax', 'pack_', 'mesh_loss',
                'adam', 'conv3x3', 'h2_', 'maxpool', 'avgpool', 'upsample', 'nchw', 'nhwc', 'gather', 'scatter', 'project',
                'add_', 'two_sum', 'colsum', 'ln_', 'nearest', 'softmax', 'presplit')
        names = [short(r['Kernel_Name']) for r in step]
        tot = 0
        print('launches that are not kernels of this package:')
        for i, r in enumerate(step):
            n = names[i]
            if n.startswith(ours):
                continue
            d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
            tot += d
            prev = next((names[j] for j in range(i - 1, -1, -1) if names[j].startswith(ours)), '-')
            nxt = next((names[j] for j in range(i + 1, len(step)) if names[j].startswith(ours)), '-')
            print('  #%4d %6.1f us grid %8s wg %4s  %-58s  after %-28s before %s'
                  % (i, d, r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', '?')),
                     n[:58], prev[:28], nxt[:28]))
        print('  total %.3f ms' % (tot / 1e3))


if __name__ == '__main__':
    main()

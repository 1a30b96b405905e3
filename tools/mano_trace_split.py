#!/usr/bin/env python
"""Per-kernel durations of the MANO kernels in a rocprofv3 kernel trace, split by grid size (the bench runs 128 and 4096 hands
in one process: the averages of --stats mix them)."""
import csv
import sys
from collections import defaultdict

rows = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        n = r['Kernel_Name']
        if 'mano' not in n:
            continue
        short = n.split('(')[0].split('::')[-1]
        grid = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X'])), int(r['Grid_Size_Y']) // max(1, int(r['Workgroup_Size_Y'])))
        rows[(short, grid)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3)
for (k, g), v in sorted(rows.items()):
    v.sort()
    print('%-32s grid %-12s n %4d  median %8.1f us  min %8.1f' % (k, g, len(v), v[len(v) // 2], v[0]))

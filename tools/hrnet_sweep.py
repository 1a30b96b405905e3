#!/usr/bin/env python
"""tools/tile_sweep.py on the convolution shapes of HRNet-W32's branches at B = 32 (BASELINE configs[3])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tile_sweep as T  # noqa: E402

T.B = 32
LAYERS = [(64, 32, 32, 3), (32, 64, 64, 3), (16, 128, 128, 3), (8, 256, 256, 3),          # the four branches' basic blocks
          (64, 64, 64, 3), (64, 64, 256, 1), (64, 256, 64, 1),                            # stage-1 bottlenecks
          (32, 64, 32, 1), (16, 128, 32, 1), (16, 128, 64, 1), (8, 256, 32, 1), (8, 256, 64, 1), (8, 256, 128, 1)]   # fuse 1x1
if __name__ == '__main__':
    for L in LAYERS:
        T.fwd(*L)
    for L in LAYERS:
        T.wgrad(*L)

#!/usr/bin/env python
"""Where does the P3 GEMM's time go?  Times the 256x128 kernel and three ablations of its loader on two shapes (hipGraph of
20 launches each): 1 = same bytes as full 128-byte-line requests from a linear region, 2 = no loads after the first two
k-tiles (compute-only ceiling of the loop), 3 = the LDS-DMA instructions reading a zero page (no memory traffic)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402
from pair_sweep import time_graph  # noqa: E402

dev = torch.device('cuda:0')
B = 64
NAMES = {0: 'real', 1: 'linear full-line', 2: 'no loads', 3: 'zero page'}
for H, Cin, Cout, k in [(64, 128, 128, 3), (32, 512, 256, 1), (16, 256, 256, 3)]:
    p = (k - 1) // 2
    M, K = B * H * H, k * k * Cin
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    y = torch.empty(B, H, H, Cout, device=dev)
    xp = ops.p3_from_f32(M, Cin, x)
    w3, Kp = ops.p3_weight(w, Cin, False)
    g3 = (H, H, Cin, H, H, k, k, 1, p, p)
    fl = 2.0 * M * Cout * K / 1e6
    out = []
    for v in (0, 1, 2, 3):
        t = time_graph(lambda: ops.gemm_p3(xp, w3, y, M, Cout, K, Cin, Kp, Cout, g3, tile=0, variant=v))
        out.append('%s %7.1f us (%5.1f TF)' % (NAMES[v], t, fl / t))
    print('%3dx%-3d %4d->%-4d k%d tile 256x128 | %s' % (H, H, Cin, Cout, k, ' | '.join(out)), flush=True)

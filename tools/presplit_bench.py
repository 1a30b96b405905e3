#!/usr/bin/env python
"""Per-shape effect of pre-split weight operands (rih_gemm b_mode 2, ops.PRESPLIT) on the ResNet50 convolutions of one
B=64 step: forward GEMM with the weight converted inside the kernel (b_mode 0 / 1) vs read as pre-split bf16 planes,
with and without the presplit pass itself, and with the activation pre-split as well (a_mode 2: the GEMM converts
nothing; the activation pass is timed separately -- it would be folded into the producing BatchNorm kernel).  Every candidate = 20 launches replayed from a hipGraph.  End to end:
`RIH_PRESPLIT=1 python bench.py`."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402
from pair_sweep import time_graph  # noqa: E402
from tile_sweep import LAYERS  # noqa: E402

dev = torch.device('cuda:0')
B = 64


def one(H, Cin, Cout, k):
    p = (k - 1) // 2
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    y = torch.empty(B, H, H, Cout, device=dev)
    M, K = B * H * H, k * k * Cin
    geom = (H, H, Cin, H, H, k, k, 1, 1, p, p)
    if k == 1:
        base = lambda: ops.gemm(x, w, y, M, Cout, K, Cin, Cin, Cout, a_mode=0, b_mode=1, geom=geom, engine=1)
    else:
        wp = torch.empty(K, Cout, device=dev)

        def base():
            ops.check(ops._L().rih_pack_conv_weight(w.data_ptr(), wp.data_ptr(), Cout, Cin, k, k, Cin, 0, ops._stream()), 'pack')
            ops.gemm(x, wp, y, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, engine=1)
    planes, Kp = ops._presplit_weight(w, Cin, False)
    only = lambda: ops.gemm(x, planes, y, M, Cout, K, Cin, Kp, Cout, a_mode=0, b_mode=2, geom=geom, engine=1)

    def full():
        pl, kp = ops._presplit_weight(w, Cin, False)
        ops.gemm(x, pl, y, M, Cout, K, Cin, kp, Cout, a_mode=0, b_mode=2, geom=geom, engine=1)
    xp = ops._presplit_act(B * H * H, Cin, x)
    both = lambda: ops.gemm(xp, planes, y, M, Cout, K, Cin, Kp, Cout, a_mode=2, b_mode=2, geom=geom, engine=1)
    act = lambda: ops._presplit_act(B * H * H, Cin, x)
    t0, t1, t2, t3, t4 = time_graph(base), time_graph(only), time_graph(full), time_graph(both), time_graph(act)
    fl = 2.0 * M * Cout * K / 1e6
    print('fwd %3dx%-3d %4d->%-4d k%d | in-kernel split %7.1f us (%5.1f TF) | B pre-split %7.1f us (%5.1f TF), with its pass'
          ' %7.1f us (%+5.1f%%) | A and B pre-split %7.1f us (%5.1f TF), activation pass %6.1f us'
          % (H, H, Cin, Cout, k, t0, fl / t0, t1, fl / t1, t2, 100 * (t2 / t0 - 1), t3, fl / t3, t4), flush=True)
    return t0, t2


if __name__ == '__main__':
    tot0 = tot2 = 0.0
    for L in LAYERS:
        if L[1] % 32 == 0 and L[2] > 32:
            a, b = one(*L)
            tot0 += a
            tot2 += b
    print('sum over the listed shapes: %.1f us -> %.1f us (%+.1f%%)' % (tot0, tot2, 100 * (tot2 / tot0 - 1)))

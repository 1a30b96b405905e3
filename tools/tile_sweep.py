#!/usr/bin/env python
"""Sweep the rih_gemm tile choice over the convolution shapes of the ResNet50 variant at B=64 (forward / data-gradient
as implicit GEMM, weight gradient as split-K) and print, per shape, every tile's time next to the planner's pick."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402
from gemm_bench import time_launch  # noqa: E402

dev = torch.device('cuda:0')
B = 64
ENG = int(os.environ.get('RIH_SWEEP_ENGINE', '1'))      # 2: engine 2 with operand bounds by rih_absmax


def _bound(t):
    a = torch.zeros(2048, device=dev)
    ops.check(ops._L().rih_absmax(t.data_ptr(), t.numel(), a.data_ptr(), ops._stream()), 'rih_absmax')
    return a

# (H, Cin, Cout, k) stride-1 layers (count per step); the planner sees forward (Cin->Cout) and data-gradient (Cout->Cin)
LAYERS = [(64, 64, 64, 1), (64, 64, 64, 3), (64, 64, 256, 1), (64, 256, 64, 1), (64, 256, 128, 1), (32, 128, 128, 3),
          (32, 128, 512, 1), (32, 512, 128, 1), (32, 512, 256, 1), (16, 256, 256, 3), (16, 256, 1024, 1),
          (16, 1024, 256, 1), (16, 1024, 512, 1), (8, 512, 512, 3), (8, 512, 2048, 1), (8, 2048, 512, 1),
          (8, 2048, 128, 1), (16, 128, 128, 3), (32, 128, 128, 3), (64, 128, 128, 3), (8, 256, 256, 1),
          (16, 1280, 256, 1), (32, 768, 256, 1), (64, 512, 256, 1)]


def fwd(H, Cin, Cout, k):
    p = (k - 1) // 2
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(k * k * Cin, Cout, device=dev)
    y = torch.empty(B, H, H, Cout, device=dev)
    M, K = B * H * H, k * k * Cin
    geom = (H, H, Cin, H, H, k, k, 1, 1, p, p)
    pick, sk = ops.plan_gemm(M, Cout, K, 1, ENG)
    kw = dict(amax_a=_bound(x), amax_b=_bound(w)) if ENG == 2 else {}
    res = {}
    for t in (0, 1, 2, 4):
        try:
            res[t] = time_launch(lambda: ops.gemm(x, w, y, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, tile=t, engine=ENG, **kw), 10)
        except RuntimeError:
            res[t] = float('inf')       # the 256x128 kernel has preconditions (Cin % 32, N % 4) and is an experiment variant
    us_plan = time_launch(lambda: ops.gemm(x, w, y, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, engine=ENG, **kw), 10)
    best = min(res, key=res.get)
    flag = '' if us_plan <= 1.05 * res[best] else '   <-- planner loses %.0f%%' % (100 * (us_plan / res[best] - 1))
    print('fwd  %3dx%-3d %4d->%-4d k%d | M%-7d N%-5d K%-5d | t0 %7.1f t1 %7.1f t2 %7.1f t4 %7.1f | plan t%d sk%d %7.1f us%s'
          % (H, H, Cin, Cout, k, M, Cout, K, res[0], res[1], res[2], res[4], pick, sk, us_plan, flag), flush=True)


def wgrad(H, Cin, Cout, k):
    p = (k - 1) // 2
    x = torch.randn(B, H, H, Cin, device=dev)
    dy = torch.randn(B, H, H, Cout, device=dev)
    dw = torch.empty(Cout, Cin, k, k, device=dev)
    geom = (H, H, Cin, H, H, k, k, 1, 1, p, p)
    M = B * H * H
    us_plan = time_launch(lambda: ops._wgrad(x, dy, dw, M, k * k * Cin, Cout, Cin, Cout, geom, Cin, k * k, Cin), 10)
    out = []
    for t in (0, 2):
        bm, bn = ops._TILE_MN[t]
        tiles = -(-(k * k * Cin) // bm) * -(-Cout // bn)
        for target in (256, 512, 1024):
            skk = max(1, min(target // tiles, -(-M // 128)))
            kc = -(-(-(-M // skk)) // 32) * 32
            skk = -(-M // kc)
            part = torch.empty(skk, k * k * Cin, Cout, device=dev)
            us = time_launch(lambda: (ops.gemm(x, dy, part, k * k * Cin, Cout, M, Cin, Cout, Cout, a_mode=1, b_mode=0, splitk=skk,
                                               kchunk=kc, sCsplit=k * k * Cin * Cout, geom=geom, tile=t, engine=1) if skk > 1 else
                                      ops.gemm(x, dy, part, k * k * Cin, Cout, M, Cin, Cout, Cout, a_mode=1, b_mode=0, geom=geom, tile=t, engine=1),
                                      ops.check(ops._L().rih_splitk_reduce(part.data_ptr(), skk, k * k * Cin, Cout, dw.data_ptr(), Cin, k * k, Cin, 0,
                                                                            ops._stream()), 'reduce')), 10)
            out.append((us, 't%d sk%d' % (t, skk)))
    best = min(out)
    flag = '' if us_plan <= 1.05 * best[0] else '   <-- planner loses %.0f%%' % (100 * (us_plan / best[0] - 1))
    print('wgrad %3dx%-3d %4d->%-4d k%d | plan %7.1f us | best %7.1f (%s) | %s%s'
          % (H, H, Cin, Cout, k, us_plan, best[0], best[1], ' '.join('%s=%.0f' % (n, u) for u, n in out), flag), flush=True)


if __name__ == '__main__':
    for L in LAYERS:
        fwd(*L)
        if L[1] != L[2]:
            fwd(L[0], L[2], L[1], L[3])       # the data gradient of a stride-1 conv has the channels swapped
    if os.environ.get('RIH_SWEEP_WGRAD', '1') == '1':
        for L in LAYERS:
            wgrad(*L)

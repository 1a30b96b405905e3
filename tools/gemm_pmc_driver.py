#!/usr/bin/env python
"""Launch plan for a rocprofv3 --pmc pass over the PRODUCTION split-engine GEMM on the twelve shapes that take the most time in
a training step (profiles/r03/gemm_profile_e2.json): forward / data-gradient type launches with the planner's tile and split-K,
weight gradients with the split rule of ops._wgrad_now.  Per shape REP launches; the plan goes out as JSON on the last line and
tools/gemm_pmc.sh joins it with the counter CSV by dispatch order (clock, MFMA-busy, wait fractions per shape: is a shape held
by the power cap -- low clock at high MFMA-busy -- or by stalls -- full clock at low MFMA-busy?)."""
import json
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from renderih_amd import ops  # noqa: E402
from gemm_pmc_driver_shapes import SHAPES  # noqa: E402

dev = torch.device('cuda:0')
B, REP = 64, 10
ENG = int(os.environ.get('RIH_PMC_ENGINE', '1'))        # 2: the three-product fp16 engine (operand bounds by rih_absmax)


def amax(t):
    a = torch.zeros(2048, device=dev)      # a bound block (include/renderih_amd.h: rih_absmax)
    ops.check(ops._L().rih_absmax(t.data_ptr(), t.numel(), a.data_ptr(), ops._stream()), 'rih_absmax')
    return a


plan = []
for kind, H, Cin, Cout, k in SHAPES:
    p = (k - 1) // 2
    M, K = B * H * H, k * k * Cin
    x = torch.randn(B, H, H, Cin, device=dev)
    geom = (H, H, Cin, H, H, k, k, 1, 1, p, p)
    if kind == 'fwd':
        wp = torch.randn(K, Cout, device=dev) / K ** 0.5
        tile, sk = ops.plan_gemm(M, Cout, K, 1, 1)
        kw = dict(amax_a=amax(x), amax_b=amax(wp)) if ENG == 2 else {}
        Bop, ldb, bm = wp, Cout, 0
        if sk > 1:
            kc = -(-(-(-K // sk)) // 32) * 32
            sk = -(-K // kc)
            part = torch.empty(sk, M, Cout, device=dev)
            fn = lambda: ops.gemm(x, Bop, part, M, Cout, K, Cin, ldb, Cout, a_mode=0, b_mode=bm, geom=geom, tile=tile, splitk=sk,
                                  kchunk=kc, sCsplit=M * Cout, engine=ENG, **kw)
        else:
            y = torch.empty(B, H, H, Cout, device=dev)
            fn = lambda: ops.gemm(x, Bop, y, M, Cout, K, Cin, ldb, Cout, a_mode=0, b_mode=bm, geom=geom, tile=tile, engine=ENG, **kw)
        label = 'fwd   %2dx%-2d %4d->%-4d k%d | M%-6d N%-4d K%-5d | tile %d sk %d' % (H, H, Cin, Cout, k, M, Cout, K, tile, sk)
    else:
        dy = torch.randn(B, H, H, Cout, device=dev)
        Mp = K
        small = (-(-Mp // 128)) * (-(-Cout // 128)) <= 4 and M < 16384
        tile = 2 if (Cout <= 64 or Mp <= 64 or small) else 0
        bm, bn = ops._TILE_MN[tile]
        tiles = (-(-Mp // bm)) * (-(-Cout // bn))
        sk = max(1, min((256 if tiles == 1 else 512) // max(tiles, 1), -(-M // 128)))
        kc = -(-(-(-M // sk)) // 32) * 32
        sk = -(-M // kc)
        part = torch.empty(sk, Mp, Cout, device=dev)
        kw = dict(amax_a=amax(x), amax_b=amax(dy)) if ENG == 2 else {}
        fn = lambda: ops.gemm(x, dy, part, Mp, Cout, M, Cin, Cout, Cout, a_mode=1, b_mode=0, splitk=sk, kchunk=kc,
                              sCsplit=Mp * Cout, geom=geom, tile=tile, engine=ENG, **kw)
        label = 'wgrad %2dx%-2d %4d->%-4d k%d | M%-6d N%-4d K%-5d | tile %d sk %d' % (H, H, Cin, Cout, k, Mp, Cout, M, tile, sk)
    for _ in range(REP):
        fn()
    torch.cuda.synchronize()
    plan.append({'label': label, 'count': REP, 'flop': 2.0 * M * Cout * K})
print(json.dumps(plan))

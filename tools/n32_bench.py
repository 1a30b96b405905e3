#!/usr/bin/env python
"""HRNet-W32's 32-channel branch: 3x3 32->32 convolutions at 64x64, B = 32 (N = 32 GEMMs).  The planner sends N <= 32 to the
128x32 tile on the native f32 MFMA (engine 0); is a half-empty 64x64 split-engine tile faster?  Forward-type GEMM and the
weight gradient, 20 launches per hipGraph."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops
from pair_sweep import time_graph
dev = torch.device('cuda:0')
B = 32
for H, Cin, Cout, k in [(64, 32, 32, 3), (64, 32, 32, 1), (64, 64, 32, 1), (64, 32, 64, 3), (128, 4, 64, 3)]:
    p = (k - 1) // 2
    cpad = max(Cin, 4)
    x = torch.randn(B, H, H, cpad, device=dev)
    wp = torch.randn(k * k * cpad, Cout, device=dev)
    y = torch.empty(B, H, H, Cout, device=dev)
    M, K = B * H * H, k * k * cpad
    geom = (H, H, cpad, H, H, k, k, 1, 1, p, p)
    fl = 2.0 * M * Cout * K / 1e6
    res = {}
    for tile, eng in ((3, 0), (2, 1), (1, 1), (2, 0)):
        if tile == 3 and Cout > 32:
            continue
        try:
            t = time_graph(lambda: ops.gemm(x, wp, y, M, Cout, K, cpad, Cout, Cout, a_mode=0, b_mode=0, geom=geom, tile=tile, engine=eng))
            res['t%d e%d' % (tile, eng)] = t
        except Exception as e:
            res['t%d e%d' % (tile, eng)] = float('nan')
    print('fwd %dx%d %d->%d k%d M%d N%d K%d | ' % (H, H, Cin, Cout, k, M, Cout, K) + ' | '.join('%s %7.1f us %5.1f TF' % (n, t, fl / t) for n, t in res.items()), flush=True)
    # weight gradient through the planner vs forced 64x64 split tile
    dy = torch.randn(B, H, H, Cout, device=dev)
    dw = torch.empty(Cout, cpad, k, k, device=dev)
    t_plan = time_graph(lambda: ops._wgrad(x, dy, dw, M, k * k * cpad, Cout, cpad, Cout, geom, cpad, k * k, cpad))
    print('   wgrad planner %7.1f us %5.1f TF' % (t_plan, fl / t_plan), flush=True)
    # explicit alternatives for the weight gradient: tile (engine) x number of K-slices
    Mp = k * k * cpad
    for tile, eng in ((3, 0), (2, 1), (2, 0)):
        if tile == 3 and Cout > 32:
            continue
        out = []
        for target in (256, 512, 1024):
            bm, bn = ops._TILE_MN[tile]
            tiles = -(-Mp // bm) * -(-Cout // bn)
            sk = max(1, min(target // tiles, -(-M // 128)))
            kc = -(-(-(-M // sk)) // 32) * 32
            sk = -(-M // kc)
            part = torch.empty(sk, Mp, Cout, device=dev)

            def run():
                ops.gemm(x, dy, part, Mp, Cout, M, cpad, Cout, Cout, a_mode=1, b_mode=0, splitk=sk, kchunk=kc, sCsplit=Mp * Cout,
                         geom=geom, tile=tile, engine=eng)
                ops.check(ops._L().rih_splitk_reduce(part.data_ptr(), sk, Mp, Cout, dw.data_ptr(), cpad, k * k, cpad, 0,
                                                     ops._stream()), 'reduce')
            try:
                out.append('sk%d %6.1f us' % (sk, time_graph(run)))
            except Exception as e:      # noqa: BLE001
                out.append('sk%d failed' % sk)
        print('   wgrad t%d e%d: %s' % (tile, eng, ' | '.join(out)), flush=True)

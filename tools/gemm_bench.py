#!/usr/bin/env python
"""Micro-benchmark of rih_gemm on the shapes that dominate a training step (from profiles/*/gemm_profile*.json):
every tile configuration per shape, HIP-event timed over a batch of launches.  Run on the GPU box:
    python tools/gemm_bench.py > gpurun_out/gemm_bench.log"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def time_launch(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000.0      # us


def err_vs_fp64(got, A2d, B2d):
    """max |got - fp64| / max|fp64| and the same for a plain fp32 torch matmul (rocBLAS) as the yardstick."""
    ref = A2d.double() @ B2d.double()
    e = float((got.double().view_as(ref) - ref).abs().max() / ref.abs().max())
    e32 = float(((A2d @ B2d).double() - ref).abs().max() / ref.abs().max())
    return e, e32


ENGINES = (0, 1)


def conv_fwd(N, H, W, Cin, Cout, k, s, p, tiles=(0, 1, 2)):
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(N, H, W, Cin, device=dev)
    wp = torch.randn(k * k * Cin, Cout, device=dev)
    y = torch.empty(N, Ho, Wo, Cout, device=dev)
    M, K = N * Ho * Wo, k * k * Cin
    geom = (H, W, Cin, Ho, Wo, k, k, s, 1, p, p)
    out = []
    for t in tiles:
        for e in ENGINES:
            us = time_launch(lambda: ops.gemm(x, wp, y, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, tile=t, engine=e))
            out.append('t%de%d %7.1fus %6.1fTF' % (t, e, us, 2.0 * M * Cout * K / us / 1e6))
    if k == 1 and s == 1:
        for e in ENGINES:
            ops.gemm(x, wp, y, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, tile=tiles[-1], engine=e)
            out.append('t%de%d err %.1e (fp32 blas %.1e)' % ((tiles[-1], e) + err_vs_fp64(y, x.view(M, K), wp)))
    else:   # conv: compare the last tile's result with the first tile's (same engine)
        y0 = torch.empty_like(y)
        ops.gemm(x, wp, y0, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, tile=tiles[0], engine=ENGINES[-1])
        ops.gemm(x, wp, y, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, tile=tiles[-1], engine=ENGINES[-1])
        out.append('t%d vs t%d maxdiff %.1e' % (tiles[-1], tiles[0], float((y - y0).abs().max() / y0.abs().max())))
    print('conv fwd  N%d %dx%d Cin%d Cout%d k%d s%d | M%d N%d K%d | %s' % (N, H, W, Cin, Cout, k, s, M, Cout, K, '  '.join(out)), flush=True)


def wgrad(N, H, W, Cin, Cout, k, s, p, splits=(None,), tiles=(0, 2)):
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(N, H, W, Cin, device=dev)
    dy = torch.randn(N, Ho, Wo, Cout, device=dev)
    Kpix, Mrows = N * Ho * Wo, k * k * Cin
    geom = (H, W, Cin, Ho, Wo, k, k, s, 1, p, p)
    out = []
    res = []
    for t in tiles:
        bm, bn = ops._TILE_MN[t]
        ntiles = -(-Mrows // bm) * -(-Cout // bn)
        for target in (256, 512, 1024, 2048):
            sk = max(1, min(target // max(ntiles, 1), -(-Kpix // 128)))
            kc = -(-(-(-Kpix // sk)) // 32) * 32
            sk = -(-Kpix // kc)
            part = torch.empty(sk, Mrows, Cout, device=dev)
            for e in ENGINES:
                us = time_launch(lambda: ops.gemm(x, dy, part, Mrows, Cout, Kpix, Cin, Cout, Cout, a_mode=1, b_mode=0,
                                                  splitk=sk, kchunk=kc, sCsplit=Mrows * Cout, geom=geom, tile=t, engine=e) if sk > 1 else
                                 ops.gemm(x, dy, part, Mrows, Cout, Kpix, Cin, Cout, Cout, a_mode=1, b_mode=0, geom=geom, tile=t, engine=e))
                out.append('t%de%d sk%3d %6.1fus %5.1fTF' % (t, e, sk, us, 2.0 * Mrows * Cout * Kpix / us / 1e6))
        res.append(part.sum(0))
    if len(res) > 1:
        out.append('t%d vs t%d maxdiff %.1e' % (tiles[-1], tiles[0], float((res[-1] - res[0]).abs().max() / res[0].abs().max())))
    print('wgrad N%d %dx%d Cin%d Cout%d k%d s%d | M%d N%d K%d | %s' % (N, H, W, Cin, Cout, k, s, Mrows, Cout, Kpix, ' | '.join(out)), flush=True)


def linear_fwd(M, K, Nf, tiles=(0, 1, 2, 3)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(Nf, K, device=dev)
    y = torch.empty(M, Nf, device=dev)
    out = []
    for t in tiles:
        bm, bn = ops._TILE_MN[t]
        if bn < 64 and Nf > 64:
            continue
        for e in (ENGINES if t != 3 else (0,)):
            us = time_launch(lambda: ops.gemm(x, w, y, M, Nf, K, K, K, Nf, a_mode=0, b_mode=1, tile=t, engine=e))
            out.append('t%de%d %6.1fus %5.1fTF' % (t, e, us, 2.0 * M * Nf * K / us / 1e6))
    for e in ENGINES:
        ops.gemm(x, w, y, M, Nf, K, K, K, Nf, a_mode=0, b_mode=1, tile=tiles[-1], engine=e)
        out.append('t%de%d err %.1e (fp32 blas %.1e)' % ((tiles[-1], e) + err_vs_fp64(y, x, w.t())))
    print('linear M%d K%d N%d | %s' % (M, K, Nf, '  '.join(out)), flush=True)


if __name__ == '__main__':
    B = 64
    if '--e1' in sys.argv:          # split-engine iteration set
        globals()['ENGINES'] = (1,)
        conv_fwd(B, 64, 64, 128, 128, 3, 1, 1, tiles=(0, 4))
        conv_fwd(B, 32, 32, 128, 128, 3, 1, 1, tiles=(0, 4))
        conv_fwd(B, 16, 16, 256, 256, 3, 1, 1, tiles=(0, 1, 4))
        conv_fwd(B, 8, 8, 512, 512, 3, 1, 1, tiles=(1, 2, 4))
        conv_fwd(B, 64, 64, 512, 256, 1, 1, 0, tiles=(0, 4))
        conv_fwd(B, 64, 64, 64, 256, 1, 1, 0, tiles=(0, 4))
        conv_fwd(B, 16, 16, 1024, 256, 1, 1, 0, tiles=(0, 4))
        wgrad(B, 32, 32, 128, 128, 3, 1, 1, tiles=(0, 4))
        wgrad(B, 16, 16, 256, 256, 3, 1, 1, tiles=(0, 4))
        wgrad(B, 16, 16, 1024, 256, 1, 1, 0, tiles=(0, 4))
        linear_fwd(8128, 256, 768, tiles=(0, 2, 4))
        sys.exit(0)
    if '--small' in sys.argv:
        globals()['ENGINES'] = (1,)
        for M, K, Nf in ((16128, 64, 64), (16128, 64, 128), (8064, 128, 128), (4032, 256, 256), (4032, 512, 256),
                         (8128, 256, 768), (20224, 64, 192)):
            linear_fwd(M, K, Nf, tiles=(2,))
        wgrad(B * 63, 1, 1, 256, 256, 1, 1, 0, tiles=(2,))
        wgrad(B * 252, 1, 1, 64, 64, 1, 1, 0, tiles=(2,))
        sys.exit(0)
    if '--pmc4' in sys.argv:
        globals()['ENGINES'] = (1,)
        conv_fwd(B, 64, 64, 128, 128, 3, 1, 1, tiles=(0, 4))
        wgrad(B, 32, 32, 128, 128, 3, 1, 1, tiles=(0,))
        sys.exit(0)
    if '--quick' in sys.argv:       # a few launches for PMC collection
        conv_fwd(B, 64, 64, 128, 128, 3, 1, 1, tiles=(0,))
        conv_fwd(B, 16, 16, 1024, 256, 1, 1, 0, tiles=(0,))
        wgrad(B, 32, 32, 128, 128, 3, 1, 1, tiles=(0,))
        sys.exit(0)
    conv_fwd(B, 64, 64, 128, 128, 3, 1, 1)        # aux decoder 64x64 stage
    conv_fwd(B, 64, 64, 64, 64, 3, 1, 1)          # layer1 3x3
    conv_fwd(B, 32, 32, 128, 128, 3, 1, 1)
    conv_fwd(B, 16, 16, 256, 256, 3, 1, 1)
    conv_fwd(B, 8, 8, 512, 512, 3, 1, 1)
    conv_fwd(B, 64, 64, 256, 64, 1, 1, 0)
    conv_fwd(B, 64, 64, 64, 256, 1, 1, 0)
    conv_fwd(B, 16, 16, 1024, 256, 1, 1, 0)
    conv_fwd(B, 8, 8, 2048, 512, 1, 1, 0)
    conv_fwd(B, 64, 64, 512, 256, 1, 1, 0)
    wgrad(B, 32, 32, 128, 128, 3, 1, 1)
    wgrad(B, 64, 64, 64, 64, 3, 1, 1)
    wgrad(B, 16, 16, 256, 256, 3, 1, 1)
    wgrad(B, 8, 8, 512, 512, 3, 1, 1)
    wgrad(B, 16, 16, 1024, 256, 1, 1, 0)
    wgrad(B * 63, 1, 1, 256, 256, 1, 1, 0)        # decoder linear weight grad
    wgrad(B * 252, 1, 1, 64, 64, 1, 1, 0)
    for M, K, Nf in ((4032, 256, 256), (4032, 512, 256), (8064, 128, 128), (16128, 64, 64), (8128, 256, 768),
                     (20224, 64, 192), (4096, 256, 768)):
        linear_fwd(M, K, Nf)

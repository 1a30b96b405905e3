#!/usr/bin/env python
"""Engine 2 (fp32 on three fp16 MFMA products) against engine 1 (six bf16 products) and rocBLAS fp32 on the twelve shapes that
take the most time in a training step (tools/gemm_pmc_driver.py, profiles/r03/gemm_pmc_table_top12.txt): HIP-event time and TF
per engine at the production tile / split-K, and the maximum error against an fp64 product of the same operands
(max |got - fp64| / max |fp64|) next to the same figure of a plain fp32 torch.matmul (rocBLAS / hipBLASLt) -- the yardstick the
round-3 verdict sets: engine 2 counts as fp32 arithmetic only where its error does not exceed the library's.  Operand
magnitudes are what a training step sees: activations after BatchNorm + ReLU, He-scaled weights, gradients of the order 1e-4
with a log-normal spread over pixels and channels (weight-gradient shapes).  Run on the GPU box:
    python tools/e2_bench.py > gpurun_out/e2_bench.log"""
import os
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from renderih_amd import ops  # noqa: E402
from gemm_pmc_driver_shapes import SHAPES  # noqa: E402

dev = torch.device('cuda:0')
B = 64
BE = 8          # images of the error check (fp64 im2col products)


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


def amax(t):
    a = torch.zeros(2048, device=dev)      # a bound block (include/renderih_amd.h: rih_absmax)
    ops.check(ops._L().rih_absmax(t.data_ptr(), t.numel(), a.data_ptr(), ops._stream()), 'rih_absmax')
    return a


def im2col(x, k, p):
    """[n, H, W, C] -> [n*H*W, k*k*C] with k ordered (kh, kw, c) like rih_gemm's gather (stride 1)."""
    n, H, W, C = x.shape
    xp = F.pad(x, (0, 0, p, p, p, p))
    cols = [xp[:, kh:kh + H, kw:kw + W, :] for kh in range(k) for kw in range(k)]
    return torch.cat(cols, dim=-1).reshape(n * H * W, k * k * C)


def errs(got, A2d, B2d):
    ref = A2d.double() @ B2d.double()
    sc = ref.abs().max()
    return float((got.double().reshape(ref.shape) - ref).abs().max() / sc), float(((A2d @ B2d).double() - ref).abs().max() / sc)


print('device', torch.cuda.get_device_name(0), flush=True)
tot = {1: 0.0, 2: 0.0}
for kind, H, Cin, Cout, k in SHAPES:
    p = (k - 1) // 2
    M, K = B * H * H, k * k * Cin
    torch.manual_seed(H * 131 + Cin + Cout + k)
    x = torch.relu(torch.randn(B, H, H, Cin, device=dev) * 1.3 + 0.2)             # BatchNorm + ReLU output
    geom = (H, H, Cin, H, H, k, k, 1, 1, p, p)
    res = {}
    if kind == 'fwd':
        wp = torch.randn(K, Cout, device=dev) * (2.0 / K) ** 0.5
        tile, sk = ops.plan_gemm(M, Cout, K, 1, 1)
        ax, aw = amax(x), amax(wp)
        for e in (1, 2):
            kw = dict(amax_a=ax, amax_b=aw) if e == 2 else {}
            if sk > 1:
                kc = -(-(-(-K // sk)) // 32) * 32
                skk = -(-K // kc)
                part = torch.empty(skk, M, Cout, device=dev)
                fn = lambda: ops.gemm(x, wp, part, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, tile=tile,
                                      splitk=skk, kchunk=kc, sCsplit=M * Cout, engine=e, **kw)
            else:
                y = torch.empty(B, H, H, Cout, device=dev)
                fn = lambda: ops.gemm(x, wp, y, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, tile=tile, engine=e, **kw)
            us = time_launch(fn)
            # error on the first BE images in the PRODUCTION configuration: same tile, same split-K chunk (the partial slabs
            # summed in fp32 in slab order, as rih_splitk_finish does)
            Me = BE * H * H
            if sk > 1:
                pe = torch.empty(skk, Me, Cout, device=dev)
                ops.gemm(x[:BE].contiguous(), wp, pe, Me, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, tile=tile,
                         splitk=skk, kchunk=kc, sCsplit=Me * Cout, engine=e, **kw)
                ye = pe[0].clone()
                for i_ in range(1, skk):
                    ye += pe[i_]
            else:
                ye = torch.empty(BE, H, H, Cout, device=dev)
                ops.gemm(x[:BE].contiguous(), wp, ye, Me, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0,
                         geom=geom, tile=tile, engine=e, **kw)
            er, e32 = errs(ye, im2col(x[:BE], k, p), wp)
            res[e] = (us, er, e32)
        label = 'fwd   %2dx%-2d %4d->%-4d k%d | M%-6d N%-4d K%-5d | tile %d sk %d' % (H, H, Cin, Cout, k, M, Cout, K, tile, sk)
    else:
        # gradient magnitudes: ~1e-4 with a log-normal spread over pixels and channels
        dy = torch.randn(B, H, H, Cout, device=dev) * 1e-4 * torch.exp(torch.randn(B, H, H, 1, device=dev) * 1.5) \
            * torch.exp(torch.randn(1, 1, 1, Cout, device=dev))
        Mp = K
        small = (-(-Mp // 128)) * (-(-Cout // 128)) <= 4 and M < 16384
        tile = 2 if (Cout <= 64 or Mp <= 64 or small) else 0
        bm, bn = ops._TILE_MN[tile]
        tiles = (-(-Mp // bm)) * (-(-Cout // bn))
        sk = max(1, min((256 if tiles == 1 else 512) // max(tiles, 1), -(-M // 128)))
        kc = -(-(-(-M // sk)) // 32) * 32
        sk = -(-M // kc)
        part = torch.empty(sk, Mp, Cout, device=dev)
        ax, ady = amax(x), amax(dy)
        for e in (1, 2):
            kw = dict(amax_a=ax, amax_b=ady) if e == 2 else {}
            fn = lambda: ops.gemm(x, dy, part, Mp, Cout, M, Cin, Cout, Cout, a_mode=1, b_mode=0, splitk=sk, kchunk=kc,
                                  sCsplit=Mp * Cout, geom=geom, tile=tile, engine=e, **kw)
            us = time_launch(fn)
            # error on the first BE images in the PRODUCTION configuration: the same k-chunk per split-K slice as the timed launch,
            # the partial slabs summed in fp32 in slab order (rih_splitk_reduce's order)
            Me = BE * H * H
            ske = -(-Me // kc)
            pe = torch.empty(ske, Mp, Cout, device=dev)
            if ske > 1:
                ops.gemm(x[:BE].contiguous(), dy[:BE].contiguous(), pe, Mp, Cout, Me, Cin, Cout, Cout, a_mode=1, b_mode=0,
                         splitk=ske, kchunk=kc, sCsplit=Mp * Cout, geom=geom, tile=tile, engine=e, **kw)
            else:
                ops.gemm(x[:BE].contiguous(), dy[:BE].contiguous(), pe, Mp, Cout, Me, Cin, Cout, Cout, a_mode=1, b_mode=0,
                         geom=geom, tile=tile, engine=e, **kw)
            ge = pe[0].clone()
            for i_ in range(1, ske):
                ge += pe[i_]
            er, e32 = errs(ge, im2col(x[:BE], k, p).t().contiguous(), dy[:BE].reshape(Me, Cout))
            res[e] = (us, er, e32)
        label = 'wgrad %2dx%-2d %4d->%-4d k%d | M%-6d N%-4d K%-5d | tile %d sk %d' % (H, H, Cin, Cout, k, Mp, Cout, M, tile, sk)
    fl = 2.0 * M * Cout * K
    for e in (1, 2):
        tot[e] += res[e][0]
    print('%s | e1 %7.1f us %6.1f TF err %.2e | e2 %7.1f us %6.1f TF err %.2e | x%.2f | fp32 blas err %.2e | e2 <= blas: %s' % (
        label, res[1][0], fl / res[1][0] / 1e6, res[1][1], res[2][0], fl / res[2][0] / 1e6, res[2][1],
        res[1][0] / res[2][0], res[2][2], res[2][1] <= res[2][2]), flush=True)
print('sum over the twelve shapes: e1 %.1f us, e2 %.1f us (x%.2f)' % (tot[1], tot[2], tot[1] / tot[2]))

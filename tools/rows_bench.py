#!/usr/bin/env python
"""The long-K plain-row GEMM (csrc/rih_conv3.hip rows_kernel, ops.ROWS) against rih_gemm's tiled kernels (with the planner's own tile and
split-K) on the 1x1 shapes of the ResNet50 step at B = 64 with K >= 256: forward with the BatchNorm statistics epilogue, data gradient
with and without a residual; HIP-event time per launch in interleaved rounds, TF/s of algorithmic FLOPs, error against fp64 on a slice.
    python tools/rows_bench.py [--hrnet]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
# (M, K, N, what, residual, statistics, launches per step)      B = 64 ResNet50: 64x64 -> M 262144, 32x32 -> 65536, 16x16 -> 16384, 8x8 -> 4096
CASES = [
    (262144, 256, 64, 'fwd conv1 L1', False, True, 2), (262144, 256, 64, 'dgrad conv3 L1', False, False, 3),
    (262144, 256, 128, 'fwd conv1 L2.0', False, True, 1), (262144, 512, 256, 'dgrad ds L2 + skip', True, False, 1),
    (65536, 512, 128, 'fwd conv1 L2', False, True, 3), (65536, 512, 128, 'dgrad conv3 L2', False, False, 4),
    (65536, 512, 256, 'fwd conv1 L3.0', False, True, 1), (65536, 256, 512, 'dgrad conv1 L3.0 + skip', True, False, 1),
    (16384, 1024, 256, 'fwd conv1 L3', False, True, 5), (16384, 1024, 256, 'dgrad conv3 L3', False, False, 6),
    (16384, 256, 1024, 'fwd conv3 L3', False, True, 6), (16384, 256, 1024, 'dgrad conv1 L3 + skip', True, False, 5),
    (16384, 1024, 512, 'fwd conv1 L4.0', False, True, 1), (16384, 512, 1024, 'dgrad conv1 L4.0 + skip', True, False, 1),
    (4096, 2048, 512, 'fwd conv1 L4', False, True, 2), (4096, 2048, 512, 'dgrad conv3 L4', False, False, 3),
    (4096, 512, 2048, 'fwd conv3 L4', False, True, 3), (4096, 512, 2048, 'dgrad conv1 L4 + skip', True, False, 2),
    (4096, 2048, 128, 'fwd aux 2048->128', False, True, 2),
    (65536, 128, 512, 'fwd conv3 L2 (panel today)', False, True, 4), (262144, 64, 256, 'fwd conv3 L1 (panel today)', False, True, 4),
]
HRNET = [   # B = 32: 64x64 -> 131072, 32x32 -> 32768, 16x16 -> 8192, 8x8 -> 2048
    (131072, 256, 64, 'hr layer1 conv1', False, True, 3), (131072, 64, 256, 'hr layer1 conv3', False, True, 4),
    (8192, 128, 128, 'hr fuse 1x1', False, True, 4), (2048, 256, 256, 'hr fuse 1x1 8x8', False, True, 4),
]
ROUNDS, ITERS = 5, 10


def timed(fn):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(ITERS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / ITERS * 1000.0


def main():
    assert ops.ENGINE == 2
    print('device', torch.cuda.get_device_name(0))
    tot = {True: 0.0, False: 0.0}
    cases = HRNET if '--hrnet' in sys.argv else CASES
    for M, K, N, what, res, stats, per_step in cases:
        torch.manual_seed(M % 1000 + K + N)
        fwd = what.startswith('fwd') or what.startswith('hr')
        a = (torch.relu(torch.randn(M, K, device=dev) * 1.3 + 0.2) if fwd else
             torch.randn(M, K, device=dev) * 1e-4 * torch.exp(torch.randn(M, 1, device=dev)))
        # the OIHW parameter: forward n = co (N), k = ci (K); data gradient n = ci (N), k = co (K)
        w = (torch.randn(N, K, 1, 1, device=dev) if fwd else torch.randn(K, N, 1, 1, device=dev)) * (2.0 / K) ** 0.5
        R = torch.randn(M, N, device=dev) * 1e-4 if res else None
        c = {p: torch.empty(M, N, device=dev) for p in (True, False)}
        ba, bw = ops.bound_of(a), ops.bound_of(w)
        ops._h2_weight(w, (K if fwd else N), not fwd)       # (per call without a PackCache: converted outside the timed region below)
        pc = ops.PackCache()

        def rows():
            h = ops.StatsHolder() if stats else None
            ops.rows_gemm(a, w, c[True], M, N, K, K, N, not fwd, stats=h, R=R, ldr=N, ba=ba, bw=bw)

        def tiled():
            h = ops.StatsHolder() if stats else None
            if fwd:
                ops.gemm(a, w, c[False], M, N, K, K, K, N, a_mode=0, b_mode=1, stats=h, amax_a=ba, amax_b=bw)
            else:
                ops.gemm(a, w, c[False], M, N, K, K, N, N, a_mode=0, b_mode=0, R=R, ldr=N, amax_a=ba, amax_b=bw)
        saved = ops._PACK
        ops._PACK = pc          # the H2 planes come out of the cache, as inside a training step (one rih_h2_multi per step)
        try:
            rows()
            pc.refresh()
            for f in (rows, tiled):
                for _ in range(3):
                    f()
            torch.cuda.synchronize()
            tp, tt = [], []
            for _ in range(ROUNDS):
                tp.append(timed(rows))
                tt.append(timed(tiled))
        finally:
            ops._PACK = saved
        mp, mt = sorted(tp)[ROUNDS // 2], sorted(tt)[ROUNDS // 2]
        fl = 2.0 * M * N * K
        Wm = (w.view(N, K) if fwd else w.view(K, N).t()).double()
        n = 4096
        ref = a[:n].double() @ Wm.t() + (R[:n].double() if res else 0)
        sc = float(ref.abs().max())
        e = {p: float((c[p][:n].double() - ref).abs().max()) / sc for p in (True, False)}
        print('%-28s M %7d K %4d N %4d | rows %7.1f us %5.0f TF/s | tiled %7.1f us %5.0f TF/s | x%.2f | err vs fp64 %.2e / %.2e'
              % (what, M, K, N, mp, fl / mp / 1e6, mt, fl / mt / 1e6, mt / mp, e[True], e[False]), flush=True)
        tot[True] += mp * per_step
        tot[False] += mt * per_step
    print('per training step (launch counts of the step): rows %.0f us, tiled %.0f us' % (tot[True], tot[False]))


if __name__ == '__main__':
    main()

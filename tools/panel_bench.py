#!/usr/bin/env python
"""The short-K streaming GEMM (csrc/rih_conv3.hip panel_kernel, ops.PANEL) against rih_gemm's tiled kernels on the 1x1 shapes of the
ResNet50 step at B = 64 with K = 64 / 128: forward with the BatchNorm statistics epilogue, data gradient with and without a
residual; HIP-event time per launch in interleaved rounds, GB/s of algorithmic bytes (A + C (+ R)), error against fp64 on a slice.
    python tools/panel_bench.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = 64
# (H = W, K, N, what, residual, statistics, launches per step)
CASES = [(64, 64, 256, 'fwd conv3 L1', False, True, 4), (64, 64, 64, 'fwd conv1 L1.0', False, True, 1),
         (32, 128, 512, 'fwd conv3 L2', False, True, 4),
         (64, 64, 256, 'dgrad conv1 L1 + skip', True, False, 2), (64, 64, 64, 'dgrad conv1 L1.0', False, False, 1),
         (32, 128, 512, 'dgrad conv1 L2 + skip', True, False, 3), (64, 128, 256, 'dgrad conv1 L2.0 + skip', True, False, 1)]
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
    for H, K, N, what, res, stats, per_step in CASES:
        M = B * H * H
        torch.manual_seed(H + K + N)
        fwd = what.startswith('fwd')
        a = (torch.relu(torch.randn(M, K, device=dev) * 1.3 + 0.2) if fwd else
             torch.randn(M, K, device=dev) * 1e-4 * torch.exp(torch.randn(M, 1, device=dev)))
        # the OIHW parameter: forward n = co (N), k = ci (K); data gradient n = ci (N), k = co (K)
        w = (torch.randn(N, K, 1, 1, device=dev) if fwd else torch.randn(K, N, 1, 1, device=dev)) * (2.0 / K) ** 0.5
        R = torch.randn(M, N, device=dev) * 1e-4 if res else None
        c = {p: torch.empty(M, N, device=dev) for p in (True, False)}
        ba, bw = ops.bound_of(a), ops.bound_of(w)

        def panel():
            h = ops.StatsHolder() if stats else None
            ops.panel_gemm(a, w, c[True], M, N, K, K, N, not fwd, stats=h, R=R, ldr=N, ba=ba, bw=bw)

        def tiled():
            h = ops.StatsHolder() if stats else None
            if fwd:
                ops.gemm(a, w, c[False], M, N, K, K, K, N, a_mode=0, b_mode=1, stats=h, amax_a=ba, amax_b=bw)
            else:
                ops.gemm(a, w, c[False], M, N, K, K, N, N, a_mode=0, b_mode=0, R=R, ldr=N, amax_a=ba, amax_b=bw)
        for f in (panel, tiled):
            for _ in range(3):
                f()
        torch.cuda.synchronize()
        tp, tt = [], []
        for _ in range(ROUNDS):
            tp.append(timed(panel))
            tt.append(timed(tiled))
        mp, mt = sorted(tp)[ROUNDS // 2], sorted(tt)[ROUNDS // 2]
        nbytes = 4.0 * M * (K + N + (N if res else 0))
        Wm = (w.view(N, K) if fwd else w.view(K, N).t()).double()
        n = 4096
        ref = a[:n].double() @ Wm.t() + (R[:n].double() if res else 0)
        sc = float(ref.abs().max())
        e = {p: float((c[p][:n].double() - ref).abs().max()) / sc for p in (True, False)}
        print('%-26s M %7d K %3d N %3d | panel %7.1f us %5.0f GB/s | tiled %7.1f us %5.0f GB/s | x%.2f | err vs fp64 %.2e / %.2e'
              % (what, M, K, N, mp, nbytes / mp / 1e3, mt, nbytes / mt / 1e3, mt / mp, e[True], e[False]), flush=True)
        tot[True] += mp * per_step
        tot[False] += mt * per_step
    print('per training step (launch counts of ResNet50, B = 64): panel %.0f us, tiled %.0f us' % (tot[True], tot[False]))


if __name__ == '__main__':
    main()

#!/usr/bin/env python
"""Round 6: (tile, split-K) candidates for the few-tile, long-K convolutions that stay on rih_gemm's tiled kernels -- HRNet's 8 x 8
branch (3x3 256 -> 256 at B = 32: M 2048, N 256, K 2304; 50 launches per step) and ResNet50's layer4 (3x3 512 -> 512 at B = 64: M 4096,
N 512, K 4608) -- through ops.conv2d with ops.plan_gemm overridden: forward including the split-K finishing pass, HIP-event time.
    python tools/r6_plan_sweep.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000.0


def main():
    real = ops.plan_gemm
    for B, H, C, Cout in ((32, 8, 256, 256), (64, 8, 512, 512), (32, 8, 256, 128)):
        x = torch.relu(torch.randn(B, H, H, C, device=dev))
        w = torch.randn(Cout, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
        ops.set_bound(x, ops.bound_of(x)) if hasattr(ops, 'set_bound') else None
        M, K = B * H * H, 9 * C
        ref = None
        print('3x3 %dx%d %d->%d B %d: M %d N %d K %d; planner: %s' % (H, H, C, Cout, B, M, Cout, K, real(M, Cout, K, 1, 2)))
        for tile, sk in ((None, None), (0, 1), (0, 3), (0, 5), (0, 9), (1, 2), (1, 3), (1, 5), (1, 9), (2, 1), (2, 2), (2, 3), (2, 5), (2, 9)):
            if tile is None:
                ops.plan_gemm = real
            else:
                ops.plan_gemm = (lambda t, s: (lambda M_, N_, K_, batch=1, engine=None: (t, s) if (M_, N_, K_) == (M, Cout, K) else real(M_, N_, K_, batch, engine)))(tile, sk)
            with torch.no_grad():
                y = ops.conv2d(x, w, None, stride=1, pad=1)
                us = timed(lambda: ops.conv2d(x, w, None, stride=1, pad=1))
            if ref is None:
                ref = y
            err = float((y - ref).abs().max() / ref.abs().max())
            print('   tile %s sk %s: %7.1f us  %6.1f TF/s   (max diff to the planner\'s result %.1e)' % (tile, sk, us, 2.0 * M * Cout * K / us / 1e6, err))
    ops.plan_gemm = real


if __name__ == '__main__':
    main()

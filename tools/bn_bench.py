#!/usr/bin/env python
"""What the BatchNorm kernels reach of the chip's streaming bandwidth, per shape of the ResNet50 step at B = 64: rih_bn_apply
(+ residual, ReLU pattern, max|y|) and rih_bn_bwd (reduction pass + finisher + apply) in GB/s of algorithmic bytes, next to a plain
torch copy (1 read + 1 write) and torch add (2 reads + 1 write) of the same tensors -- the yardstick for "how far from a copy is
this kernel", which a whole-step average (launch ramps of 126 calls included) cannot tell.  python tools/bn_bench.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = 64
SHAPES = [(64, 256, True), (64, 64, False), (32, 512, True), (32, 128, False), (16, 1024, True), (16, 256, False), (8, 2048, True),
          (8, 512, False)]      # (H = W, channels, with residual): bn3 + skip / bn1-bn2 of each layer


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000.0


def main():
    L = ops._L()
    print('device', torch.cuda.get_device_name(0))
    for H, Cc, res in SHAPES:
        rows = B * H * H
        x = torch.randn(rows, Cc, device=dev)
        r = torch.randn(rows, Cc, device=dev) if res else None
        y, dx = torch.empty_like(x), torch.empty_like(x)
        dres = torch.empty_like(x) if res else None
        dy = torch.randn(rows, Cc, device=dev) * 1e-3
        mean, invstd = x.mean(0), 1.0 / (x.var(0, unbiased=False) + 1e-5).sqrt()
        g, b = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
        mask = torch.empty(rows * Cc // 4, device=dev, dtype=torch.uint8)
        amax = torch.zeros(2048, device=dev)
        dg, db = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
        ws = torch.empty(int(L.rih_bn_ws_floats(rows, Cc)), device=dev)
        s = ops._stream()
        nb = x.numel() * 4.0

        def apply():
            ops.check(L.rih_bn_apply(x.data_ptr(), mean.data_ptr(), invstd.data_ptr(), g.data_ptr(), b.data_ptr(), ops._p(r),
                                     y.data_ptr(), rows, Cc, 1, mask.data_ptr(), amax.data_ptr(), s), 'bn_apply')

        def bwd():
            ops.check(L.rih_bn_bwd(dy.data_ptr(), x.data_ptr(), 0, mean.data_ptr(), invstd.data_ptr(), g.data_ptr(), dx.data_ptr(),
                                   ops._p(dres), dg.data_ptr(), db.data_ptr(), rows, Cc, 1, 0, ws.data_ptr(), mask.data_ptr(),
                                   amax.data_ptr(), s), 'bn_bwd')
        t_apply, t_bwd = timed(apply), timed(bwd)
        t_copy = timed(lambda: y.copy_(x))
        t_add = timed(lambda: torch.add(x, dy, out=y))
        b_apply = nb * (2 + (1 if res else 0)) + nb / 16
        b_bwd = nb * (2 * 2 + 1 + (1 if res else 0)) + 2 * nb / 16
        print('%3dx%-3d C %4d %s | apply %7.1f us %5.0f GB/s | bwd (3 launches) %7.1f us %5.0f GB/s | copy %7.1f us %5.0f GB/s | add %7.1f us %5.0f GB/s'
              % (H, H, Cc, 'res' if res else '   ', t_apply, b_apply / t_apply / 1e3, t_bwd, b_bwd / t_bwd / 1e3, t_copy,
                 2 * nb / t_copy / 1e3, t_add, 3 * nb / t_add / 1e3), flush=True)


if __name__ == '__main__':
    main()

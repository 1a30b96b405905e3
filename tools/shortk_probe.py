#!/usr/bin/env python
"""The short-reduction 1x1 convolutions of layer1..3 (K = 64..256) run at 2.4 TB/s of algorithmic traffic:
what do the tiles / engines / rocBLAS / a plain streaming copy of the same bytes do on these shapes?
    python tools/shortk_probe.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402
from gemm_bench import time_launch  # noqa: E402

dev = torch.device('cuda:0')
SHAPES = [(262144, 256, 64), (262144, 64, 256), (65536, 512, 128), (65536, 128, 512), (16384, 1024, 256), (262144, 256, 128)]


def main():
    for M, N, K in SHAPES:
        x = torch.randn(M, K, device=dev)
        w = torch.randn(K, N, device=dev)           # b_mode 0: [K][N]
        wt = w.t().contiguous()                     # b_mode 1: [N][K]
        y = torch.empty(M, N, device=dev)
        r = torch.randn(M, N, device=dev)
        geom = (1, 1, K, 1, 1, 1, 1, 1, 1, 0, 0)
        byts = 4.0 * (M * K + K * N + M * N)
        out = []
        for t in (0, 1, 2, 4):
            for e in (1,):
                try:
                    us = time_launch(lambda: ops.gemm(x, w, y, M, N, K, K, N, N, a_mode=0, b_mode=0, geom=geom, tile=t, engine=e), 20)
                    out.append('t%d %6.1f' % (t, us))
                except RuntimeError:
                    out.append('t%d    n/a' % t)
        us1 = time_launch(lambda: ops.gemm(x, wt, y, M, N, K, K, K, N, a_mode=0, b_mode=1, geom=geom, engine=1), 20)
        us_res = time_launch(lambda: ops.gemm(x, w, y, M, N, K, K, N, N, a_mode=0, b_mode=0, geom=geom, engine=1, R=r, ldr=N), 20)
        us_e0 = time_launch(lambda: ops.gemm(x, w, y, M, N, K, K, N, N, a_mode=0, b_mode=0, geom=geom, engine=0), 20)
        us_blas = time_launch(lambda: torch.mm(x, w, out=y), 20)
        us_copy = time_launch(lambda: y.copy_(r), 20)                  # 8 B per output element: read + write M x N
        us_fill = time_launch(lambda: y.fill_(1.0), 20)                # write only
        pick, sk = ops.plan_gemm(M, N, K, 1, 1)
        print('M%-7d N%-5d K%-4d %5.0f MB | %s | plan t%d | b_mode1 %6.1f | +residual %6.1f | engine0 %6.1f | rocBLAS %6.1f | '
              'copy MxN %6.1f (%.2f TB/s) | fill MxN %6.1f (%.2f TB/s) | floor@5TB/s %5.1f us'
              % (M, N, K, byts / 1e6, ' '.join(out), pick, us1, us_res, us_e0, us_blas, us_copy, 8.0 * M * N / us_copy / 1e6,
                 us_fill, 4.0 * M * N / us_fill / 1e6, byts / 5e6), flush=True)


if __name__ == '__main__':
    main()

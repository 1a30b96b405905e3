#!/usr/bin/env python
"""Driver for the PMC pass over rows_kernel (tools/r5_pmc.sh r6pmc_rows 'rows_kernel|gemm_split_kernel' tools/rows_pmc_driver.py):
the three 1x1 shapes the round-5 verdict names (16x16 1024 -> 256, 256 -> 1024, 32x32 512 -> 128 at B = 64) and the largest one
(64x64 512 -> 256 data gradient), each launched 12 times on rows_kernel and 12 times on the tiled kernel."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from renderih_amd import ops  # noqa: E402

d = torch.device('cuda:0')
for M, K, N in ((16384, 1024, 256), (16384, 256, 1024), (65536, 512, 128), (262144, 512, 256), (262144, 256, 64)):
    torch.manual_seed(K + N)
    a = torch.relu(torch.randn(M, K, device=d) * 1.3 + 0.2)
    w = torch.randn(N, K, 1, 1, device=d) * (2.0 / K) ** 0.5
    c = torch.empty(M, N, device=d)
    ba, bw = ops.bound_of(a), ops.bound_of(w)
    pc = ops.PackCache()
    ops._PACK = pc
    ops.rows_gemm(a, w, c, M, N, K, K, N, False, stats=ops.StatsHolder(), ba=ba, bw=bw)
    pc.refresh()
    for _ in range(12):
        ops.rows_gemm(a, w, c, M, N, K, K, N, False, stats=ops.StatsHolder(), ba=ba, bw=bw)
    ops._PACK = None
    for _ in range(12):
        ops.gemm(a, w, c, M, N, K, K, K, N, a_mode=0, b_mode=1, stats=ops.StatsHolder(), amax_a=ba, amax_b=bw)
    torch.cuda.synchronize()
print('done')

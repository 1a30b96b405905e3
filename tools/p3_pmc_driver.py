#!/usr/bin/env python
"""Launch plan for a rocprofv3 --pmc pass over the split-engine GEMMs: per shape 10 launches each of (a) the in-kernel-split
kernel of rih_gemm, (b) the P3 kernel on random operands, (c) the P3 kernel on zero-filled operands (same instruction
stream, no data toggling: separates a power / clock limit from a structural one).  Prints the plan as JSON on the last line;
tools/p3_pmc.sh joins it with the counter CSV by dispatch order."""
import json
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B, REP = 64, 10
SHAPES = [(64, 128, 128, 3, 1), (16, 256, 256, 3, 1), (32, 512, 256, 1, 0), (64, 64, 256, 1, 0)]
plan = []
for H, Cin, Cout, k, tile in SHAPES:
    p = (k - 1) // 2
    M, K = B * H * H, k * k * Cin
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    y = torch.empty(B, H, H, Cout, device=dev)
    geom = (H, H, Cin, H, H, k, k, 1, 1, p, p)
    if k == 1:
        base = lambda: ops.gemm(x, w, y, M, Cout, K, Cin, Cin, Cout, a_mode=0, b_mode=1, geom=geom, engine=1)
    else:
        wp = torch.empty(K, Cout, device=dev)
        ops.check(ops._L().rih_pack_conv_weight(w.data_ptr(), wp.data_ptr(), Cout, Cin, k, k, Cin, 0, ops._stream()), 'pack')
        base = lambda: ops.gemm(x, wp, y, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, engine=1)
    xp = ops.p3_from_f32(M, Cin, x)
    w3, Kp = ops.p3_weight(w, Cin, False)
    xz, wz = torch.zeros_like(xp), torch.zeros_like(w3)
    g3 = (H, H, Cin, H, H, k, k, 1, p, p)
    torch.cuda.synchronize()
    tag = '%dx%d %d->%d k%d' % (H, H, Cin, Cout, k)
    fl = 2.0 * M * Cout * K
    for name, fn in (('split-in-kernel', base),
                     ('p3 t%d random' % tile, lambda: ops.gemm_p3(xp, w3, y, M, Cout, K, Cin, Kp, Cout, g3, tile=tile)),
                     ('p3 t%d zeros' % tile, lambda: ops.gemm_p3(xz, wz, y, M, Cout, K, Cin, Kp, Cout, g3, tile=tile))):
        for _ in range(REP):
            fn()
        torch.cuda.synchronize()
        plan.append({'label': tag + ' | ' + name, 'count': REP, 'flop': fl})
print(json.dumps(plan))

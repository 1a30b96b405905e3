#!/usr/bin/env python
"""Round 6: stress of rows_kernel for run-to-run reproducibility -- every shape of the step, 30 launches each on the same operands
(interleaved with an unrelated kernel that rewrites a scratch buffer), outputs compared bitwise with the first launch and against the
tiled kernel; then the same with the operand PRODUCED by a kernel right in front of every launch."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from renderih_amd import ops  # noqa: E402

d = torch.device('cuda:0')
REPS = int(os.environ.get('REPS', '30'))
SHAPES = [(262144, 256, 64, True), (262144, 256, 64, False), (262144, 256, 128, True), (65536, 256, 512, False), (65536, 512, 128, True),
          (16384, 1024, 256, True), (16384, 256, 1024, True), (262144, 512, 256, False)]
scratch = torch.empty(64 << 20, device=d)
bad = 0
for M, K, N, fwd in SHAPES:
    torch.manual_seed(M % 977 + K + N)
    src = torch.randn(M, K, device=d)
    w = (torch.randn(N, K, 1, 1, device=d) if fwd else torch.randn(K, N, 1, 1, device=d)) * (2.0 / K) ** 0.5
    R = None if fwd else torch.randn(M, N, device=d) * 0.1
    ref = torch.empty(M, N, device=d)
    a0 = torch.relu(src * 1.3 + 0.2)
    ba, bw = ops.bound_of(a0), ops.bound_of(w)
    if fwd:
        ops.gemm(a0, w, ref, M, N, K, K, K, N, a_mode=0, b_mode=1, amax_a=ba, amax_b=bw)
    else:
        ops.gemm(a0, w, ref, M, N, K, K, N, N, a_mode=0, b_mode=0, R=R, ldr=N, amax_a=ba, amax_b=bw)
    first, ndiff, worst = None, 0, 0.0
    for rep in range(REPS):
        a = torch.relu(src * 1.3 + 0.2) if rep % 2 else a0          # odd launches: the operand comes fresh out of a producer kernel
        c = torch.empty(M, N, device=d)
        h = ops.StatsHolder() if fwd else None
        ok = ops.rows_gemm(a, w, c, M, N, K, K, N, not fwd, stats=h, R=R, ldr=N, ba=ba, bw=bw)
        assert ok
        scratch.normal_()                                           # unrelated traffic behind the launch
        if first is None:
            first = c.clone()
            worst = float((c - ref).abs().max() / ref.abs().max())
        elif not torch.equal(c, first):
            ndiff += 1
            dd = (c - first).abs()
            rows = torch.nonzero(dd.amax(1) > 0).flatten()
            print('   launch %d differs: %d rows, first rows %s, max |diff| %.3g, non-finite %d' % (
                rep, rows.numel(), rows[:6].tolist(), float(dd.max()), int((~torch.isfinite(c)).sum())), flush=True)
    torch.cuda.synchronize()
    bad += ndiff
    print('M %7d K %4d N %4d %s: %d of the repeat launches differ; rows vs tiled %.2e of max' % (M, K, N, 'fwd' if fwd else 'dgrad', ndiff, worst),
          flush=True)
print('TOTAL differing launches:', bad)

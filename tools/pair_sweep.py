#!/usr/bin/env python
"""Tile / split-K sweep over the hands-paired decoder GEMMs (batch 2) of one B=64 training step.  Shapes are read from
the per-launch profile bench.py dumped (profiles/r01/gemm_profile_v15.json); every candidate is timed as 20 launches
replayed from a hipGraph (the Python launch path would hide 15-30 us kernels)."""
import json
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from renderih_amd import ops  # noqa: E402

dev = torch.device('cuda:0')


def time_graph(fn, iters=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1000.0


def shapes():
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r01', 'gemm_profile_v15.json')))
    ix = {c: i for i, c in enumerate(d['columns'])}
    seen = {}
    for r in d['rows']:
        if r[ix['batch']] != 2:
            continue
        k = (r[ix['M']], r[ix['N']], r[ix['K']], r[ix['a_mode']], r[ix['b_mode']])
        seen.setdefault(k, [0, r[ix['tile']], r[ix['splitk']], 0.0])
        seen[k][0] += 1
        seen[k][3] += r[ix['us']]
    return seen


def fwd_like(M, N, K, b_mode, info):
    A = torch.randn(2, M, K, device=dev)
    W = torch.randn(2, N, K, device=dev) if b_mode == 1 else torch.randn(2, K, N, device=dev)
    Cc = torch.empty(2, M, N, device=dev)
    ldb = K if b_mode == 1 else N
    res = {}
    for t in (0, 1, 2):
        res[t] = time_graph(lambda: ops.gemm(A, W, Cc, M, N, K, K, ldb, N, a_mode=0, b_mode=b_mode, nb1=2, sA=(M * K, 0),
                                             sB=(N * K, 0), sC=(M * N, 0), tile=t, engine=1))
    pick, _ = ops.plan_gemm(M, N, K, 2, 1)
    best = min(res, key=res.get)
    flag = '' if res[pick] <= 1.05 * res[best] else '   <-- planner loses %.0f%%' % (100 * (res[pick] / res[best] - 1))
    print('a0 b%d M%-6d N%-4d K%-5d x%d | t0 %6.1f t1 %6.1f t2 %6.1f | plan t%d (bench %5.1f us)%s'
          % (b_mode, M, N, K, info[0], res[0], res[1], res[2], pick, info[3] / info[0], flag), flush=True)
    return info[0] * (res[pick] - res[best])


def wgrad_like(Mp, N, Kpix, info):
    Mrows = Mp - 4
    x = torch.randn(2, Kpix, Mrows, device=dev)
    dy = torch.randn(2, Kpix, N, device=dev)
    dw = torch.empty(2, N, Mrows, device=dev)
    db = torch.empty(2, N, device=dev)
    geom = (1, 1, Mrows, 1, 1, 1, 1, 1, 1, 0, 0)
    us_plan = time_graph(lambda: ops._wgrad(x, dy, dw, Kpix, Mrows, N, Mrows, N, geom, Mrows, 1, Mrows, db=db, nb=2,
                                            sx=Kpix * Mrows, sdy=Kpix * N))
    out = []
    for t in (0, 2):
        bm, bn = ops._TILE_MN[t]
        tiles = -(-Mp // bm) * -(-N // bn) * 2
        for target in (256, 512, 1024, 2048):
            sk = max(1, min(target // tiles, -(-Kpix // 128)))
            kc = -(-(-(-Kpix // sk)) // 32) * 32
            sk = -(-Kpix // kc)
            part = torch.empty(2, sk, Mp, N, device=dev)

            def run():
                ops.gemm(x, dy, part, Mp, N, Kpix, Mrows, N, N, a_mode=1, b_mode=0, splitk=sk, kchunk=kc,
                         sCsplit=Mp * N, geom=geom, tile=t, engine=1, ones_row=Mrows, nb1=2, sA=(Kpix * Mrows, 0),
                         sB=(Kpix * N, 0), sC=(sk * Mp * N, 0))
                ops.check(ops._L().rih_splitk_reduce_bias_batched(part.data_ptr(), sk, Mp, Mrows, N, dw.data_ptr(), Mrows,
                                                                   1, Mrows, 0, db.data_ptr(), 2, sk * Mp * N, N * Mrows, N,
                                                                   ops._stream()), 'reduce')
            if sk > 1:
                out.append((time_graph(run), 't%d sk%d' % (t, sk)))
    best = min(out)
    flag = '' if us_plan <= 1.05 * best[0] else '   <-- planner loses %.0f%%' % (100 * (us_plan / best[0] - 1))
    print('wgrad Mp%-4d N%-4d Kpix%-6d x%d | plan %6.1f us | best %6.1f (%s) | %s%s'
          % (Mp, N, Kpix, info[0], us_plan, best[0], best[1], ' '.join('%s=%.0f' % (n, u) for u, n in out), flag), flush=True)
    return info[0] * (us_plan - best[0])


if __name__ == '__main__':
    lost = 0.0
    sh = shapes()
    for (M, N, K, am, bm), info in sorted(sh.items(), key=lambda kv: -kv[1][3]):
        if am == 0:
            lost += fwd_like(M, N, K, bm, info)
    print('fwd-like: planner leaves %.0f us per step on the table' % lost)
    lost = 0.0
    for (M, N, K, am, bm), info in sorted(sh.items(), key=lambda kv: -kv[1][3]):
        if am == 1 and M < 2000:
            lost += wgrad_like(M, N, K, info)
    print('wgrad: planner leaves %.0f us per step on the table' % lost)

#!/usr/bin/env python
"""Per-shape comparison of the conversion-free P3 GEMM (csrc/rih_gemm3.hip: operands pre-split in HBM, LDS-DMA staging,
one barrier per k-tile) with the in-kernel-split engine of rih_gemm on the ResNet50 convolutions of one B=64 step.
Every candidate = 20 launches replayed from a hipGraph, random operands.  Also times the standalone fp32 -> P3 pass
(the cost that moves into the BatchNorm apply kernels).
    python tools/p3_bench.py [--json out.json]"""
import argparse
import json
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import ops  # noqa: E402
from pair_sweep import time_graph  # noqa: E402
from tile_sweep import LAYERS  # noqa: E402

dev = torch.device('cuda:0')
B = 64


def one(H, Cin, Cout, k):
    p = (k - 1) // 2
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5
    y = torch.empty(B, H, H, Cout, device=dev)
    M, K = B * H * H, k * k * Cin
    geom = (H, H, Cin, H, H, k, k, 1, 1, p, p)
    if k == 1:
        base = lambda: ops.gemm(x, w, y, M, Cout, K, Cin, Cin, Cout, a_mode=0, b_mode=1, geom=geom, engine=1)
    else:
        wp = torch.empty(K, Cout, device=dev)
        ops.check(ops._L().rih_pack_conv_weight(w.data_ptr(), wp.data_ptr(), Cout, Cin, k, k, Cin, 0, ops._stream()), 'pack')
        base = lambda: ops.gemm(x, wp, y, M, Cout, K, Cin, Cout, Cout, a_mode=0, b_mode=0, geom=geom, engine=1)
    t_base = time_graph(base)
    y0 = y.clone()
    xp = ops.p3_from_f32(M, Cin, x)
    w3, Kp = ops.p3_weight(w, Cin, False)
    g3 = (H, H, Cin, H, H, k, k, 1, p, p)
    res = {}
    for t in (0, 1, 2):
        bm, bn = ops.P3_TILES[t]
        if Cout < bn and t != 2:
            continue
        res[t] = time_graph(lambda: ops.gemm_p3(xp, w3, y, M, Cout, K, Cin, Kp, Cout, g3, tile=t))
    err = float((y - y0).abs().max() / y0.abs().max())
    xs = ops.p3_from_f32(M, Cin, x, layout=1)
    ws, _ = ops.p3_weight(w, Cin, False, layout=1)
    slab = {}
    for t in res:
        slab[t] = time_graph(lambda: ops.gemm_p3(xs, ws, y, M, Cout, K, Cin, Kp, Cout, g3, tile=t, layout=1))
    err_s = float((y - y0).abs().max() / y0.abs().max())
    ts = min(slab, key=slab.get)
    st = torch.empty(M // 128, Cout, 2, device=dev)
    tb = min(res, key=res.get)
    t_stats = time_graph(lambda: ops.gemm_p3(xp, w3, y, M, Cout, K, Cin, Kp, Cout, g3, tile=tb, stats=st))
    t_cvt = time_graph(lambda: ops.p3_from_f32(M, Cin, x, out=xp))
    fl = 2.0 * M * Cout * K / 1e6
    row = {'H': H, 'Cin': Cin, 'Cout': Cout, 'k': k, 'base_us': round(t_base, 1), 'base_tf': round(fl / t_base, 1),
           'p3_us': {str(t): round(v, 1) for t, v in res.items()}, 'p3_best_tile': tb, 'p3_best_tf': round(fl / res[tb], 1),
           'p3_with_stats_us': round(t_stats, 1), 'cvt_us': round(t_cvt, 1), 'max_rel_diff_vs_base': err,
           'p3s_us': {str(t): round(v, 1) for t, v in slab.items()}, 'p3s_best_tile': ts, 'p3s_best_tf': round(fl / slab[ts], 1),
           'p3s_max_rel_diff_vs_base': err_s}
    print('fwd %3dx%-3d %4d->%-4d k%d | split-in-kernel %7.1f us (%5.1f TF) | P3 %s | best t%d %5.1f TF (%.2fx), +stats %7.1f us'
          ' | fp32->P3 pass %6.1f us | diff %.1e || slab-major %s | best t%d %5.1f TF (%.2fx) diff %.1e'
          % (H, H, Cin, Cout, k, t_base, fl / t_base, ' '.join('t%d %7.1f' % (t, v) for t, v in res.items()), tb, fl / res[tb],
             t_base / res[tb], t_stats, t_cvt, err, ' '.join('t%d %7.1f' % (t, v) for t, v in slab.items()), ts, fl / slab[ts],
             t_base / slab[ts], err_s), flush=True)
    return row


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--json', default=None)
    a = ap.parse_args()
    rows = []
    for L in LAYERS:
        if L[1] % 32 == 0 and L[2] >= 64:
            rows.append(one(*L))
    tb = sum(r['base_us'] for r in rows)
    tp = sum(min(r['p3_us'].values()) for r in rows)
    tsl = sum(min(r['p3s_us'].values()) for r in rows)
    print('sum over the listed shapes: in-kernel split %.1f us -> P3 %.1f us (%.2fx) -> slab-major P3S %.1f us (%.2fx)'
          % (tb, tp, tb / tp, tsl, tb / tsl))
    if a.json:
        json.dump(rows, open(a.json, 'w'), indent=1)

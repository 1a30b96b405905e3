#!/usr/bin/env python
"""MANO layer micro-benchmark (B in {128, 4096} hands, PCA-45): forward (and forward+backward) of
renderih_amd.manolayer.ManoLayer replayed from a hipGraph, timed with HIP events on the launch stream.

Roofline: the layer is HBM-bound.  Algorithmic bytes per launch = 1.46 MB of constant basis (posedirs 1.26 MB,
shapedirs 93 KB, skinning weights 50 KB, joint regressor 50 KB, template 9 KB; read once) + per hand
268 B in (root 9 + pose 45 + shape 10 + trans 3 floats) + 9588 B out (778 + 21 points x 3 floats).
    python tools/mano_bench.py [--hands 128 4096] [--iters 50] [--json out.json]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import assets                                  # noqa: E402
from renderih_amd.manolayer import ManoLayer, rodrigues_batch    # noqa: E402

BASIS_BYTES = 4 * (778 * 3 * 135 + 778 * 3 * 10 + 778 * 16 + 16 * 778 + 778 * 3)
PER_HAND_BYTES = 4 * (9 + 45 + 10 + 3) + 4 * 3 * (778 + 21)
HBM_PEAK = 8.0e12


def time_graph(fn, iters):
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
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / iters * 1e-3)
    best.sort()
    return best[len(best) // 2]


def measure(B, iters, dev, variant=0):
    from renderih_amd import manolayer
    manolayer.VARIANT = variant
    layer = ManoLayer(assets.synthetic_mano_dict('right')).to(dev)
    g = torch.Generator().manual_seed(0)
    root = rodrigues_batch(torch.randn(B, 3, generator=g)).to(dev)
    pose = (0.5 * torch.randn(B, 45, generator=g)).to(dev)
    shape = torch.randn(B, 10, generator=g).to(dev)
    trans = torch.randn(B, 3, generator=g).to(dev)

    def fwd():
        with torch.no_grad():
            return layer(root, pose, shape, trans)

    t_f = time_graph(fwd, iters)
    pose_g, shape_g, root_g = pose.clone().requires_grad_(), shape.clone().requires_grad_(), root.clone().requires_grad_()
    dv = torch.randn(B, 778, 3, device=dev)

    def fwdbwd():
        v, j = layer(root_g, pose_g, shape_g, trans)
        torch.autograd.backward([v], [dv])
        pose_g.grad = shape_g.grad = root_g.grad = None

    t_fb = time_graph(fwdbwd, max(5, iters // 5))
    alg = BASIS_BYTES + B * PER_HAND_BYTES
    return {'hands': B, 'fwd_us': round(t_f * 1e6, 2), 'fwd_hands_per_s': round(B / t_f, 0),
            'fwd_algorithmic_bytes': alg, 'fwd_GBps': round(alg / t_f / 1e9, 1), 'fwd_frac_of_8TBps': round(alg / t_f / HBM_PEAK, 4),
            'fwdbwd_us': round(t_fb * 1e6, 2), 'fwdbwd_hands_per_s': round(B / t_fb, 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hands', type=int, nargs='+', default=[128, 4096])
    ap.add_argument('--iters', type=int, default=50)
    ap.add_argument('--json', default=None)
    ap.add_argument('--variant', type=int, default=0, help='0 = fused single-launch forward (auto form), 2 / 3 = its hand-major / tile-major form, 1 = round-1 two-kernel forward')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    rows = [dict(measure(B, a.iters, dev, a.variant), variant=a.variant) for B in a.hands]
    for r in rows:
        print(json.dumps(r), flush=True)
    if a.json:
        with open(a.json, 'w') as f:
            json.dump(rows, f, indent=1)


if __name__ == '__main__':
    main()

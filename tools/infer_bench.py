#!/usr/bin/env python
"""Inference-only throughput (BASELINE configs[4] shape: batch 256, eval mode, encoder + attention decoder, then the
MANO layer on 2 x 256 hands) eagerly and replayed from a captured hipGraph.  Default fp32 (the split-bf16 MFMA engine);
--fp16 runs encoder + mid model with fp16 storage and folded BatchNorm (renderih_amd/half.py; decoder and MANO stay fp32)
and also reports the deviation of the predicted vertices from the fp32 path on the same inputs.  The whole forward is stream-ordered through the C ABI (no host sync, no allocation inside
the library), which is what makes it capturable.
    python tools/infer_bench.py [--batch 256] [--iters 20]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import assets                                  # noqa: E402
from renderih_amd.model import build_model                       # noqa: E402
from renderih_amd.manolayer import ManoLayer, rodrigues_batch    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--encoder', default='resnet50')
    ap.add_argument('--fp16', action='store_true', help='fp16-storage backbone (HandNET_GCN.use_fp16_backbone)')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    model = build_model(dropout=0.05, encoder_type=a.encoder).to(dev).eval()
    mano = {s: ManoLayer(assets.synthetic_mano_dict(s)).to(dev) for s in ('left', 'right')}
    B = a.batch
    img = torch.randn(B, 3, 256, 256, device=dev)
    dev_err = tally = None
    if a.fp16:
        with torch.no_grad():
            ref32 = model(img[:8])[0]['verts3d']
            model.use_fp16_backbone()
            got16 = model(img[:8])[0]['verts3d']
        dev_err = max(float((got16[s] - ref32[s]).abs().max() / ref32[s].abs().max()) for s in ('left', 'right'))
        from renderih_amd import half
        half.TALLY = {'flop': 0.0, 'bytes': 0.0, 'launches': 0}
        with torch.no_grad():
            model(img)
        tally, half.TALLY = half.TALLY, None
    root = rodrigues_batch(torch.randn(B, 3)).to(dev)
    pose, shape = (0.5 * torch.randn(B, 45)).to(dev), torch.randn(B, 10).to(dev)

    def forward():
        with torch.no_grad():
            out = model(img)
            hands = [mano[s](root, pose, shape) for s in ('left', 'right')]
        return out, hands

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    t_eager = timed(forward, a.iters)
    ref = forward()[0][0]['verts3d']['left'].clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        forward()                                   # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    with torch.cuda.graph(g):
        static_out = forward()
    t_graph = timed(g.replay, a.iters)
    got = static_out[0][0]['verts3d']['left']
    same = bool(torch.equal(got, ref))
    print(json.dumps({'metric': 'inference images/sec (eval forward + MANO layer), batch %d, %s' % (B, a.encoder),
                      'eager_img_s': round(B / t_eager, 1), 'hipgraph_img_s': round(B / t_graph, 1),
                      'eager_ms': round(1e3 * t_eager, 2), 'hipgraph_ms': round(1e3 * t_graph, 2),
                      'graph_output_bit_identical_to_eager': same,
                      'dtype': 'f16 backbone storage / f32 accumulate, f32 decoder' if a.fp16 else 'f32',
                      'verts_rel_dev_vs_f32': dev_err,
                      # algorithmic work of the fp16 convolutions per batch: divide by their summed rocprofv3 duration
                      # (hconv_kernel rows of the kernel stats) for TFLOP/s against the 2.5 PF dense f16 MFMA peak
                      'hconv_tflop_per_batch': round(tally['flop'] / 1e12, 3) if a.fp16 else None,
                      'hconv_gb_per_batch': round(tally['bytes'] / 1e9, 3) if a.fp16 else None,
                      'hconv_launches': tally['launches'] if a.fp16 else None}))


if __name__ == '__main__':
    main()

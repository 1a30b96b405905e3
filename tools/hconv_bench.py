#!/usr/bin/env python
"""Per-layer microbenchmark of the fp16 implicit-GEMM convolution (csrc/rih_half.hip) on the shapes of the folded ResNet50
backbone at 256x256 input: time, TFLOP/s against the 2.5 PF dense f16 MFMA peak and GB/s of algorithmic traffic against
HBM, per distinct (geometry, epilogue) layer.  The table tells which layers are MFMA- and which HBM-bound and where the
128 x 128 tile falls short -- the tuning base for the next round.
    python tools/hconv_bench.py [--batch 256] [--iters 20]"""
import argparse
import json
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from renderih_amd import half          # noqa: E402

# (name, H, Cin, Cout, k, stride, residual, order, out_f32)
LAYERS = [
    ('stem 7x7/2', 256, 8, 64, 7, 2, False, 'conv-bn', False),
    ('l1 1x1 64->64', 64, 64, 64, 1, 1, False, 'conv-bn', False),
    ('l1 3x3 64', 64, 64, 64, 3, 1, False, 'conv-bn', False),
    ('l1 1x1 64->256 +res', 64, 64, 256, 1, 1, True, 'conv-bn', False),
    ('l1 1x1 256->64', 64, 256, 64, 1, 1, False, 'conv-bn', False),
    ('l2 3x3/2 128', 64, 128, 128, 3, 2, False, 'conv-bn', False),
    ('l2 3x3 128', 32, 128, 128, 3, 1, False, 'conv-bn', False),
    ('l2 1x1 128->512 +res', 32, 128, 512, 1, 1, True, 'conv-bn', False),
    ('l2 1x1 512->128', 32, 512, 128, 1, 1, False, 'conv-bn', False),
    ('l3 3x3 256', 16, 256, 256, 3, 1, False, 'conv-bn', False),
    ('l3 1x1 256->1024 +res', 16, 256, 1024, 1, 1, True, 'conv-bn', False),
    ('l3 1x1 1024->256', 16, 1024, 256, 1, 1, False, 'conv-bn', False),
    ('l4 3x3 512', 8, 512, 512, 3, 1, False, 'conv-bn', False),
    ('l4 1x1 512->2048 +res', 8, 512, 2048, 1, 1, True, 'conv-bn', False),
    ('l4 1x1 2048->512', 8, 2048, 512, 1, 1, False, 'conv-bn', False),
    ('aux 3x3 256 @64', 64, 256, 256, 3, 1, False, 'conv-relu-bn', False),
    ('aux head 256->42 f32', 64, 256, 42, 1, 1, False, None, True),
    ('mid 1x1 768->256 f32', 64, 768, 256, 1, 1, False, 'conv-relu-bn', True),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    rows = []
    for name, H, Cin, Cout, k, stride, res, order, f32 in LAYERS:
        cin_w = 3 if Cin == 8 else Cin
        conv = nn.Conv2d(cin_w, Cout, k, stride, (k - 1) // 2, bias=(order is None)).to(dev)
        bn = nn.BatchNorm2d(Cout).to(dev).eval() if order else None
        pc = half.PackedConv(conv, bn, order, cin_pad=Cin)
        x = torch.randn(a.batch, H, H, Cin, device=dev).to(torch.float16)
        Ho = (H + 2 * ((k - 1) // 2) - k) // stride + 1
        r = torch.randn(a.batch, Ho, Ho, Cout, device=dev).to(torch.float16) if res else None
        out = torch.empty(a.batch, Ho, Ho, Cout, device=dev, dtype=torch.float32 if f32 else torch.float16)
        for _ in range(3):
            pc(x, relu=True, res=r, out=out, out_f32=f32)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            pc(x, relu=True, res=r, out=out, out_f32=f32)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        flop = 2.0 * a.batch * Ho * Ho * Cout * k * k * cin_w
        byts = 2.0 * (x.numel() + Cout * k * k * cin_w) + out.numel() * out.element_size() + (2.0 * r.numel() if res else 0)
        rows.append({'layer': name, 'ms': round(ms, 4), 'tflops': round(flop / ms / 1e9, 1), 'gbps': round(byts / ms / 1e6, 1),
                     'frac_mfma': round(flop / ms / 1e9 / 2500.0, 3), 'frac_hbm': round(byts / ms / 1e6 / 8000.0, 3)})
        print(json.dumps(rows[-1]), flush=True)
    print(json.dumps({'batch': a.batch, 'total_ms_listed': round(sum(r['ms'] for r in rows), 3)}))


if __name__ == '__main__':
    main()

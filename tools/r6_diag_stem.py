#!/usr/bin/env python
"""Round 6 diagnostic: is the stem convolution (forward rih_stem, weight gradient rih_gemm) bit-reproducible at B = 64, and which of
y / dw moves?  Usage: [RIH_STEM=0|1] python tools/r6_diag_stem.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from renderih_amd import ops  # noqa: E402

torch.manual_seed(0)
d = torch.device('cuda:0')
B = 64
img = torch.randn(B, 3, 256, 256, device=d)
w = (torch.randn(64, 3, 7, 7, device=d) * 0.08).requires_grad_(True)
gy = torch.randn(B, 128, 128, 64, device=d) * 1e-3
ys, dws = [], []
for rep in range(4):
    ops.bounds_reset()
    x = ops.nchw_to_nhwc(img, cpad=4)
    with ops.owned_bounds():
        y = ops.conv2d(x, w, None, stride=2, pad=3)
        (dw,) = torch.autograd.grad([y], [w], grad_outputs=[gy])
    torch.cuda.synchronize()
    ys.append(y.detach().clone())
    dws.append(dw.clone())
print('STEM', ops.STEM, 'y identical across 4 runs:', all(torch.equal(ys[0], t) for t in ys[1:]),
      '| dw identical:', all(torch.equal(dws[0], t) for t in dws[1:]),
      '| max |dw diff| / max|dw| %.3g' % max(float((dws[0] - t).abs().max() / dws[0].abs().max()) for t in dws[1:]))
ref = torch.nn.functional.conv2d(img.double(), w.detach().double(), stride=2, padding=3).permute(0, 2, 3, 1)
print('y vs fp64: %.3g of max' % float((ys[0].double() - ref).abs().max() / ref.abs().max()))

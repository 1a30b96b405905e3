#!/usr/bin/env python
"""Round 6 diagnostic: two identical eager steps of the B = 64 model -- which of the stem's tensors (output, output gradient, weight
gradient, the image's bound block) differ between them?  [RIH_STEM=0|1] python tools/r6_diag_model_stem.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from oracle import net_oracle  # noqa: E402
from renderih_amd import ops, testing  # noqa: E402
from renderih_amd.model import build_model  # noqa: E402

m = build_model(0.0)
m.load_state_dict(testing.deterministic_state(m.state_dict(), seed=1))
m = m.cuda().train()
img = testing.seeded_image(64, 5).cuda()
rec = []
real = ops.stem_conv


def spy(x, w, y, relu=False, stats=None, bx=None, bw=None):
    r = real(x, w, y, relu=relu, stats=stats, bx=bx, bw=bw)
    b = bx() if callable(bx) else bx
    rec[-1]['xbound'] = b.clone()
    rec[-1]['x'] = x.clone()
    return r
ops.stem_conv = spy
conv1 = m.encoder.resnet.conv1
bn1 = m.encoder.resnet.bn1
for rep in range(3):
    rec.append({})
    m.zero_grad(set_to_none=True)
    out = m(img)
    net_oracle.scalar_loss(out).backward()
    torch.cuda.synchronize()
    rec[-1]['dw'] = conv1.weight.grad.clone()
    rec[-1]['dbn'] = bn1.weight.grad.clone()
    rec[-1]['l1'] = m.encoder.resnet.layer1[0].conv1.weight.grad.clone()
for k in rec[0]:
    print('%-7s identical across steps: %s   max rel diff %.3g' % (
        k, all(torch.equal(rec[0][k], r[k]) for r in rec[1:]),
        max(float((rec[0][k] - r[k]).abs().max() / rec[0][k].abs().max().clamp_min(1e-30)) for r in rec[1:])))
if 'xbound' in rec[0]:
    print('x bound maxima per step:', [float(r['xbound'].max()) for r in rec], ' true max|x|:', float(rec[0]['x'].abs().max()))

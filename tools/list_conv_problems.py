#!/usr/bin/env python
"""List the distinct convolution / Linear problems of a model's training-mode forward (for tests/test_gpu_bench_shapes.py):
runs the forward on CPU through the numpy ABI emulation with ops.conv2d / ops.linear replaced by shape recorders that return
zeros (no arithmetic), at batch 1 -- the problem list does not depend on the batch.
    python tools/list_conv_problems.py --model hrnet32 | family_b"""
import argparse
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from abi_emulator import emulated_abi  # noqa: E402
from renderih_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='hrnet32')
    a = ap.parse_args()
    convs, lins = {}, {}

    class FakeConv(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, bias, stride, pad, relu, skip=False, grad_masked=False, stats=None):
            B, H, W, Cin = x.shape
            Cout, _, k, _ = w.shape
            Ho = (H + 2 * pad - k) // stride + 1
            key = (H, w.shape[1], Cout, k, stride, pad, bias is not None)
            convs[key] = convs.get(key, 0) + 1
            y = torch.zeros(B, Ho, Ho, Cout)
            return (y, x.view_as(x)) if skip else y

    def fake_linear(x, w, bias=None, residual=None, relu=False):
        key = (x.numel() // x.shape[-1], w.shape[1], w.shape[0], bias is not None)
        lins[key] = lins.get(key, 0) + 1
        return torch.zeros(x.shape[:-1] + (w.shape[0],))

    ops.Conv2dFn = FakeConv
    ops.conv2d = lambda x, w, bias=None, stride=1, pad=0, relu=False, grad_masked=False, stats=None: \
        FakeConv.apply(x, w, bias, stride, pad, relu, False, grad_masked, None)
    ops.conv2d_skip = lambda x, w, bias=None, stride=1, pad=0, relu=False, grad_masked=False, stats=None: \
        FakeConv.apply(x, w, bias, stride, pad, relu, True, grad_masked, None)
    ops.linear = fake_linear
    ops.GEMM_STATS = False
    with emulated_abi(), torch.no_grad():
        if a.model.startswith('hrnet'):
            from renderih_amd.encoder import HRnet_encoder, hrnet_mid
            enc = HRnet_encoder(a.model).train()
            mid = hrnet_mid(a.model).train()
            out = enc(torch.zeros(1, 3, 256, 256))
            mid(*out[3:]) if len(out) > 3 else None
        else:
            from renderih_amd.lijun import build_graph_model
            m = build_graph_model(dropout=0.0).train()
            m(torch.zeros(1, 3, 256, 256))
    print('# (H, Cin, Cout, k, stride, pad, bias): launches per forward')
    for k in sorted(convs, key=lambda t: (-t[0], t[1], t[2], t[3])):
        print('    %s,   # x%d' % (k, convs[k]))
    print('# Linear (rows at batch 1, in, out, bias)')
    for k in sorted(lins):
        print('    %s,   # x%d' % (k, lins[k]))


if __name__ == '__main__':
    main()

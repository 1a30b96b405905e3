#!/usr/bin/env python
"""Offline check (≈15 min on 8 cores, no GPU): the whole mesh decoder, forward + backward at B=1, executed through the
REAL kernels on the HIP-on-CPU harness (tests/hipcpu) and compared with the numpy ABI emulator (tests/abi_emulator.py):
≈1500 launches -- paired GEMMs with split-K weight gradients, grouped LayerNorm, Chebyshev, attention, gathers,
projection.  Round-1 result: outputs agree to 3.5e-6, every parameter gradient to 1.9e-5 (relative to its maximum)."""
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'hipcpu')]
from abi_emulator import emulated_abi  # noqa: E402
from host_kernels import host_kernels_abi  # noqa: E402
from renderih_amd import testing  # noqa: E402
from renderih_amd.model import build_model  # noqa: E402
from renderih_amd.testing import rel_err  # noqa: E402

dec = build_model(0.0).decoder
dec.load_state_dict(testing.deterministic_state(dec.state_dict(), seed=5))
dec.train()
g = torch.Generator().manual_seed(11)
x0 = torch.randn(1, dec.gf_dim, generator=g)
f0 = [torch.randn(1, s, s, 256, generator=g) * 0.5 for s in (8, 16, 32)] + [torch.zeros(1, 64, 64, 256)]


def run():
    dec.zero_grad()
    x = x0.clone().requires_grad_(True)
    fm = [t.clone().requires_grad_(True) for t in f0]
    result, params, hands, other = dec(x, fm)
    outs = {'v3d_l': result['verts3d']['left'], 'v3d_r': result['verts3d']['right'], 'v2d_l': result['verts2d']['left']}
    gen = torch.Generator().manual_seed(3)
    sum((v * torch.randn(v.shape, generator=gen)).sum() for v in outs.values()).backward()
    grads = {k: p.grad.clone() for k, p in dec.named_parameters() if p.grad is not None}
    grads['x'] = x.grad.clone()
    return {k: v.detach().clone() for k, v in outs.items()}, grads


t = time.time()
with host_kernels_abi():
    o1, g1 = run()
print('real kernels on the CPU harness: %.0f s' % (time.time() - t), flush=True)
with emulated_abi():
    o0, g0 = run()
print('outputs  :', {k: '%.2e' % rel_err(o1[k], o0[k]) for k in o0})
ranked = sorted(((rel_err(g1[k], g0[k]), k) for k in g0 if not testing.is_null_gradient(k)), reverse=True)
worst = ranked[:3]
print('gradients: worst', [('%.2e' % e, k) for e, k in worst])
if os.environ.get('DECODER_TOP'):        # more of the ranking, and how many elements of the worst tensor carry the difference
    for e, k in ranked[:int(os.environ['DECODER_TOP'])]:
        d = (g1[k] - g0[k]).abs()
        big = int((d > 0.1 * d.max()).sum())
        print('  %.2e  %-60s  elements within 10x of the largest difference: %d of %d' % (e, k, big, d.numel()))
# A hidden unit whose pre-activation is within round-off of zero can sit on different sides of its ReLU in the two evaluation
# orders: its row of the weight gradient, its bias element and the LayerNorm in front of it then differ by ~1e-3 while everything
# else agrees (round 3, default path with the flash attention kernels: one unit of layers.2.attn.ffR.fc1; with RIH_FLASH_ATTN=0
# the same draw has no such unit and the worst gradient is 7.4e-5).  Such a tensor is recognised by how FEW of its elements carry
# the difference and is held to 5e-3; every other tensor to 1e-3.
def localized(k):
    d = (g1[k] - g0[k]).abs()
    return int((d > 0.1 * d.max()).sum()) <= max(1, d.numel() // 50)


flips = {k.rsplit('.', 2)[0] for e, k in ranked if e >= 1e-3 and localized(k)}
bad = [(e, k) for e, k in ranked if e >= (5e-3 if (localized(k) or k.rsplit('.', 2)[0] in flips) else 1e-3)]
assert max(rel_err(o1[k], o0[k]) for k in o0) < 1e-4 and not bad, bad

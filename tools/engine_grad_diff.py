#!/usr/bin/env python
"""Per-parameter gradient difference between the two rih_gemm engines (same weights, same batch, dropout off)."""
import os
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import synth_batch                                   # noqa: E402
from renderih_amd import assets, ops                            # noqa: E402
from renderih_amd.model import build_model                      # noqa: E402
from renderih_amd.loss import GraphLoss, calc_loss_GCN          # noqa: E402
from renderih_amd.manolayer import ManoLayer                    # noqa: E402

dev = torch.device('cuda', 0)
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model = build_model(dropout=0.0).to(dev).train()
mano = {s: ManoLayer(assets.synthetic_mano_dict(s)) for s in ('left', 'right')}
gl = {s: GraphLoss(mano[s].J_regressor, mano[s].get_faces(), level=4, device=dev) for s in ('left', 'right')}
conv = model.decoder.converter
img, lab = synth_batch(B, dev, seed=0)
sd = {k: v.clone() for k, v in model.state_dict().items()}


def grads(engine):
    ops.ENGINE = engine
    model.load_state_dict(sd)
    model.zero_grad(set_to_none=True)
    o = model(img)
    loss, _ = calc_loss_GCN(None, 0, gl['left'], gl['right'], conv['left'], conv['right'], *o,
                            lab['v2d_l'], lab['v2d_r'], lab['v3d_l'], lab['v3d_r'], lab['root_rel'], 256)
    loss.backward()
    return float(loss), {k: p.grad.double().clone() for k, p in model.named_parameters() if p.grad is not None}


l0, g0 = grads(0)
l0b, g0b = grads(0)
l1, g1 = grads(1)
print('loss e0 %.6f e0(again) %.6f e1 %.6f' % (l0, l0b, l1))
rows = []
for k in g0:
    d = float((g1[k] - g0[k]).abs().max() / (g0[k].abs().max() + 1e-300))
    dn = float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-300))
    rows.append((d, dn, k, float(g0[k].abs().max())))
rows.sort(reverse=True)
print('worst 15 tensors by max-rel difference (engine 1 vs engine 0):')
for d, dn, k, m in rows[:15]:
    print('  %.2e (l2 %.2e)  max|g| %.3e  %s' % (d, dn, m, k))
contrib = sorted(((float((g1[k] - g0[k]).norm()), float(g0[k].norm()), k) for k in g0), reverse=True)
print('top 15 contributors to |g1-g0| (abs l2 diff, |g0|):')
for dn, n0, k in contrib[:15]:
    print('  %.3e  %.3e  rel %.2e  %s' % (dn, n0, dn / (n0 + 1e-300), k))
tot0 = torch.cat([v.flatten() for v in g0.values()])
tot1 = torch.cat([g1[k].flatten() for k in g0])
print('global: |g1-g0|/|g0| = %.3e ; cos = %.12f' % (float((tot1 - tot0).norm() / tot0.norm()),
                                                   float(torch.dot(tot0, tot1) / (tot0.norm() * tot1.norm()))))
rep = max(float((g0b[k] - g0[k]).abs().max() / (g0[k].abs().max() + 1e-300)) for k in g0)
print('engine 0 run-to-run max-rel difference: %.2e' % rep)

"""Fused HIP mesh loss (csrc/rih_loss.hip) against the reference's own core/Loss.py (golden values and gradients) and
against the torch mirror at the full batch size."""
import os
import numpy as np
import pytest
import torch

from renderih_amd import testing
from renderih_amd.testing import assert_close
from test_oracle_golden import _loss_inputs, GOLDEN

pytestmark = pytest.mark.gpu


def dev():
    """cuda:0 (tests/test_kernels_on_cpu.py re-runs these tests on the host-compiled kernels with this patched)"""
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    return torch.device('cuda:0')


PREDS = ['v3d_left', 'v3d_right', 'v2d_left', 'v2d_right', 'c3d_left', 'c3d_right', 'c2d_left', 'c2d_right']


def _dicts(t):
    result = {'verts3d': {s: t['v3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['v2d_' + s] for s in ('left', 'right')}}
    hd = [{'verts3d': {s: t['c3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['c2d_' + s] for s in ('left', 'right')}}]
    return result, hd


@pytest.mark.parametrize('epoch', [0, 60])
def test_fused_loss_matches_reference_golden(epoch):
    from renderih_amd.loss import FusedMeshLoss, calc_loss_GCN_fused
    z = np.load(os.path.join(GOLDEN, 'loss.npz'))
    t, conv, gl = _loss_inputs(z, device=dev())
    for k in PREDS:
        t[k].requires_grad_(True)
    fused = FusedMeshLoss(gl['left'], gl['right'], conv['left'], conv['right'])
    result, hd = _dicts(t)
    total, mano = calc_loss_GCN_fused(fused, epoch, result, None, hd, None, t['v2d_gt_left'], t['v2d_gt_right'],
                                      t['v3d_gt_left'], t['v3d_gt_right'], t['root_rel'])
    total.backward()
    key = 'e%d/' % epoch
    assert abs(total.item() - float(z[key + 'total'])) <= 1e-5 * abs(float(z[key + 'total']))
    for k in ('vert2d_loss', 'vert3d_loss', 'joint_loss', 'norm_loss', 'edge_loss'):
        assert abs(mano[k].item() - float(z[key + k])) <= 1e-5 * abs(float(z[key + k])) + 1e-12, k
    for k in PREDS:
        assert_close(t[k].grad, torch.from_numpy(z[key + 'grad_' + k]), 1e-4, 1e-6, 'grad ' + k)


def test_fused_loss_matches_torch_mirror_at_full_batch():
    """B = 64 (the bench batch): value and every gradient against the torch mirror on the GPU; gradients scale with the
    incoming gradient; two evaluations are bit-identical (no atomics)."""
    from renderih_amd.loss import FusedMeshLoss, calc_loss_GCN_fused, calc_loss_GCN
    z = np.load(os.path.join(GOLDEN, 'loss.npz'))
    _, conv, gl = _loss_inputs(z, device=dev())
    B = 64
    g = torch.Generator().manual_seed(7)
    t = {}
    for s in ('left', 'right'):
        t['v3d_gt_' + s] = 0.05 * torch.randn(B, 778, 3, generator=g)
        t['v2d_gt_' + s] = 256 * torch.rand(B, 778, 2, generator=g)
        t['v3d_' + s] = t['v3d_gt_' + s] + 0.5 * torch.randn(B, 778, 3, generator=g)
        t['v2d_' + s] = t['v2d_gt_' + s] + 20 * torch.randn(B, 778, 2, generator=g)
        t['c3d_' + s] = 0.6 * torch.randn(B, 252, 3, generator=g)
        t['c2d_' + s] = 256 * torch.rand(B, 252, 2, generator=g)
    t['root_rel'] = 0.05 * torch.randn(B, 3, generator=g)
    t = {k: v.to(dev()) for k, v in t.items()}
    fused = FusedMeshLoss(gl['left'], gl['right'], conv['left'], conv['right'])
    outs = []
    for mode in ('fused', 'fused', 'mirror'):
        for k in PREDS:
            t[k] = t[k].detach().requires_grad_(True)
        result, hd = _dicts(t)
        if mode == 'fused':
            total, _ = calc_loss_GCN_fused(fused, 60, result, None, hd, None, t['v2d_gt_left'], t['v2d_gt_right'],
                                           t['v3d_gt_left'], t['v3d_gt_right'], t['root_rel'])
        else:
            total, _ = calc_loss_GCN(None, 60, gl['left'], gl['right'], conv['left'], conv['right'], result, None, hd, None,
                                     t['v2d_gt_left'], t['v2d_gt_right'], t['v3d_gt_left'], t['v3d_gt_right'], t['root_rel'], 256)
        (3.0 * total).backward()
        outs.append((total.detach().clone(), {k: t[k].grad.clone() for k in PREDS}))
    assert torch.equal(outs[0][0], outs[1][0]) and all(torch.equal(outs[0][1][k], outs[1][1][k]) for k in PREDS)
    assert abs(float(outs[0][0]) - float(outs[2][0])) <= 2e-5 * abs(float(outs[2][0]))
    for k in PREDS:
        assert_close(outs[0][1][k], outs[2][1][k], 1e-4, 1e-6, 'grad ' + k)

"""CPU validation of the host-side logic (renderih_amd.ops + module tree) against PyTorch/the oracle, with the C ABI
emulated on host memory (tests/abi_emulator.py).  The kernels themselves are proven by the -m gpu tests; this pins
descriptor geometry, packing, strides, split-K plumbing, backward wiring and the module forward graph."""
import numpy as np
import pytest
import torch

import test_gpu_ops as G
from abi_emulator import emulated_abi
from renderih_amd import assets, testing
from renderih_amd.testing import assert_close


@pytest.fixture(autouse=True)
def _emulate(monkeypatch):
    monkeypatch.setattr(G, 'dev', lambda: torch.device('cpu'))
    with emulated_abi():
        yield


@pytest.mark.parametrize('case', G.CONV_CASES[:10])
def test_conv2d_host_logic(case):
    G.test_conv2d(case)


@pytest.mark.parametrize('case', G.LIN_CASES[:5] + G.LIN_CASES[6:7])
def test_linear_host_logic(case):
    G.test_linear(case)


def test_other_ops_host_logic():
    G.test_batchnorm(True, True, True, (2, 16, 16, 64))
    G.test_batchnorm(False, True, False, (3, 7, 9, 256))
    G.test_layernorm(4, 509, False, False)
    G.test_layernorm(300, 256, True, True)
    G.test_attention(2, 127, 127, 256, 4)
    G.test_attention(2, 126, 252, 128, 4)
    G.test_attention_dropout_matches_hash_mask()
    G.test_self_attention_packed(2, 190, 128, 4)
    G.test_cross_attention_packed(2, 63, 256, 4)
    G.test_add_dropout_and_bcast()
    G.test_cheby_gather_project()


def test_pool_layout_host_logic(monkeypatch):
    G.test_pool_upsample_layout()
    G.test_resample_hrnet(4, 5, 7, 32)
    G.test_resample_hrnet(2, 8, 8, 64)


def test_hrnet_host_logic_matches_oracle():
    """HRNet-W32 variant, B=1, eval mode (running-statistics BN: well conditioned), forward only: the module graph
    (transitions, fuse sums, heads, mid head) against the oracle at 1e-4."""
    from oracle import net_oracle
    from renderih_amd.model import build_model
    m = build_model(0.0, 'hrnet32')
    sd = testing.deterministic_state(m.state_dict(), seed=3)
    m.load_state_dict(sd)
    m.eval()
    img = testing.seeded_image(1, 4)
    with torch.no_grad():
        got = testing.flatten_outputs(m(img))
        graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
        want = testing.flatten_outputs(net_oracle.handnet_forward(sd, graph, img, training=False))
    for k in want:
        if not k.startswith('params.'):
            assert testing.rel_err(got[k], want[k]) < 1e-4, (k, testing.rel_err(got[k], want[k]))


def test_model_host_logic_matches_oracle():
    """Whole network, B=2, train mode, dropout 0: forward outputs and all parameter gradients.  Train-mode BN over
    <=128 samples amplifies fp32 round-off to ~1e-4, so the bar is anchored on the fp64 oracle (testing.py)."""
    from oracle import net_oracle
    from renderih_amd.model import build_model
    m = build_model(0.0)
    sd = testing.deterministic_state(m.state_dict(), seed=3)
    m.load_state_dict(sd)
    m.train()
    img = testing.seeded_image(2, 4)
    got = testing.flatten_outputs(m(img))
    graph = net_oracle.graph_from_dicts(assets.load_graph_dict('left'), assets.load_graph_dict('right'))
    w32, g32 = net_oracle.run(sd, graph, img, True, torch.float32, True)
    w64, g64 = net_oracle.run(sd, graph, img, True, torch.float64, True)
    for k in w64:
        testing.assert_fp32_equivalent(got[k], w32[k], w64[k], what=k)
    net_oracle.scalar_loss(_unflatten(got)).backward()
    from test_gpu_model import _grad_report
    params = [(k, p.grad) for k, p in m.named_parameters() if p.grad is not None]
    assert {k for k, _ in params} == set(g64.keys())
    _grad_report(params, g32, g64)
    assert int(m.encoder.resnet.bn1.num_batches_tracked) == 1


def _unflatten(f):
    sides = ('left', 'right')
    return ({'verts3d': {s: f['result.verts3d.' + s] for s in sides}, 'verts2d': {s: f['result.verts2d.' + s] for s in sides}},
            {'scale': {s: f['params.scale.' + s] for s in sides}, 'trans2d': {s: f['params.trans2d.' + s] for s in sides}},
            [{'verts3d': {s: f['hand0.verts3d.' + s] for s in sides}, 'verts2d': {s: f['hand0.verts2d.' + s] for s in sides}}],
            {'hms': f['other.hms'], 'mask': f['other.mask'], 'dense': f['other.dense']})


@pytest.mark.parametrize('epoch', [0, 60])
def test_fused_loss_host_logic_matches_reference_golden(epoch):
    """Host side of the fused mesh loss (topology tables, weights, autograd plumbing) with the kernel emulated, against
    the values and gradients of the reference's own core/Loss.py (tests/golden/loss.npz)."""
    import os
    from test_oracle_golden import _loss_inputs, GOLDEN
    from renderih_amd.loss import FusedMeshLoss, calc_loss_GCN_fused
    z = np.load(os.path.join(GOLDEN, 'loss.npz'))
    t, conv, gl = _loss_inputs(z)
    preds = ['v3d_left', 'v3d_right', 'v2d_left', 'v2d_right', 'c3d_left', 'c3d_right', 'c2d_left', 'c2d_right']
    for k in preds:
        t[k].requires_grad_(True)
    fused = FusedMeshLoss(gl['left'], gl['right'], conv['left'], conv['right'])
    result = {'verts3d': {s: t['v3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['v2d_' + s] for s in ('left', 'right')}}
    hd = [{'verts3d': {s: t['c3d_' + s] for s in ('left', 'right')}, 'verts2d': {s: t['c2d_' + s] for s in ('left', 'right')}}]
    import renderih_amd.ops as ops_mod
    orig = ops_mod._stream
    ops_mod._stream = lambda: 0
    try:
        total, mano = calc_loss_GCN_fused(fused, epoch, result, None, hd, None, t['v2d_gt_left'], t['v2d_gt_right'],
                                          t['v3d_gt_left'], t['v3d_gt_right'], t['root_rel'])
        total.backward()
    finally:
        ops_mod._stream = orig
    key = 'e%d/' % epoch
    assert abs(total.item() - float(z[key + 'total'])) <= 1e-5 * abs(float(z[key + 'total']))
    for k in ('vert2d_loss', 'vert3d_loss', 'joint_loss', 'norm_loss', 'edge_loss'):
        assert abs(mano[k].item() - float(z[key + k])) <= 1e-5 * abs(float(z[key + k])) + 1e-12, k
    for k in preds:
        assert_close(t[k].grad, torch.from_numpy(z[key + 'grad_' + k]), 1e-4, 1e-6, 'grad ' + k)
